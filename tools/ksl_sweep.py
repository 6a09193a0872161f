"""In-launch K slices (KSL) of the channels-last bf16 kernels: time the tile-starved layers of a net under (tile, slices) and check every sliced result.

  python tools/ksl_sweep.py prebuild [net] [batch]     # no GPU: hiprtc-compile every (op, tile, slices) into boda_amd/_kcache (travels to the GPU box)
  python tools/ksl_sweep.py run [net] [batch] [iters]  # GPU: median us per launch by the backend's events; each sliced output against the unsliced one (max rel diff on
                                                       # max(1, |v|)), twice (bitwise equal: the slice order of the sum is fixed)
Env: SEL=i,j,...  ops by index; MAXPEL=N  only layers with at most N output positions per image (default 196: the 14 x 14 and 7 x 7 maps)
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from boda_amd.cnn_op import OpTune, add_codegen_annotations

mode = sys.argv[1] if len(sys.argv) > 1 else "run"
net = sys.argv[2] if len(sys.argv) > 2 else "googlenet_conv"
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 64
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 12
SLICES = [int(x) for x in os.environ.get("SLICES", "1 2 4 8").split()]
IMPL_TILES = os.environ.get("IMPL_TILES", "128x128x64x2x2x2 64x128x64x1x4x2 64x64x64x2x2x2 32x128x64x1x4x2").split()
PATCH_TILES = os.environ.get("PATCH_TILES", "128x128x0x4x1x2 64x128x0x2x2x2 128x64x0x4x1x2 64x64x0x2x2x2 32x128x0x1x4x2").split()


def the_ops():
    seen = {}
    for op in bench.net_conv_ops(net, batch):
        seen.setdefault(op.to_str(), op)
    ops = list(seen.values())
    maxpel = int(os.environ.get("MAXPEL", "196"))
    idx = [i for i, op in enumerate(ops) if op.conv_geom()["OH"] * op.conv_geom()["OW"] <= maxpel]
    if os.environ.get("SEL"):
        idx = [int(i) for i in os.environ["SEL"].split(",")]
    return [(i, ops[i]) for i in idx]


def configs(op):
    g = op.conv_geom()
    patch = g["KH"] * g["KW"] >= 2 and g["SX"] == 1 and g["C"] % 8 == 0 and g["KH"] < g["H"]
    tiles = PATCH_TILES if patch else IMPL_TILES
    out = [""]
    for t in tiles:
        for s in SLICES:
            ring = "3"
            if ":" in t:     # "tile:ring" (the implicit-GEMM kernel's LDS ring depth; default 3)
                t, ring = t.split(":")
            out.append(t + (f"x{s}" if patch else f"x{s}x32x{ring}"))
    return out


def anno(op, t):
    return add_codegen_annotations(op, OpTune(hip_dtype="bf16", hip_layout="nhwc", hip_tile=t, hip_out=os.environ.get("OUT", "")))


if mode == "prebuild":
    import boda_amd.rtc as rtc
    n = bad = 0
    for i, op in the_ops():
        for t in configs(op):
            try:
                rtc.prebuild(anno(op, t), tile=t); n += 1
            except Exception as e:  # unsupported combos are skipped at run time too
                bad += 1
    print(f"prebuilt {n} specialisations ({bad} unsupported)")
    sys.exit(0)

from boda_amd.op import UnsupErr, RtErr
from boda_amd.ops_prof import OpsBackend, profile_rcg_call
from boda_amd.rtc import make_rtc
rtc = make_rtc("(be=hip)", 0); rtc.init(); be = OpsBackend(rtc)
nbad = 0
for i, op in the_ops():
    g = op.conv_geom()
    base = None
    print(f"op {i:2d} C{g['C']:4d} {g['H']:3d}x{g['W']:3d} OC{g['OC']:4d} k{g['KH']}s{g['SY']} {op.flops()/1e9:6.2f} GF", flush=True)
    for t in configs(op):
        try:
            a = anno(op, t)
            outs, prc = profile_rcg_call(be, a, 5, 0.0, iters, want_outs=True, tile=t)
            us = float(np.median(prc.all_secs[2:])) * 1e6
            o = outs["out"].astype(np.float64)
            if base is None:
                base = o
            mrd = float(np.max(np.abs(o - base) / np.maximum(1.0, np.maximum(np.abs(o), np.abs(base)))))
            outs2, _ = profile_rcg_call(be, a, 5, 0.0, 3, want_outs=True, tile=t)
            same = bool(np.array_equal(outs2["out"], outs["out"]))
            flag = "" if (mrd < (2e-3 if os.environ.get("OUT") == "f32" else 1.2e-2) and same) else "   <-- BAD"
            nbad += 1 if flag else 0
            print(f"   {t or 'auto':28s} {us:7.1f} us  [{prc.launch['cfg']:24s} grid {prc.launch['grid']:5d}]  mrd {mrd:.1e}  rerun-equal {same}{flag}", flush=True)
        except (UnsupErr, RtErr) as e:
            print(f"   {t:28s}   n/a  ({str(e)[:80]})", flush=True)
print("BAD results:", nbad)
rtc.close()
sys.exit(1 if nbad else 0)
