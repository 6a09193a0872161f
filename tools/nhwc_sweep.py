"""GPU tool: time hip_conv_nhwc on chosen layers under a list of tiles ("BIxBJxBKxWIxWJxMINWx1x32xNBUF").  python tools/nhwc_sweep.py [net] [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from boda_amd.cnn_op import OpTune, add_codegen_annotations
from boda_amd.op import UnsupErr, RtErr
from boda_amd.ops_prof import OpsBackend, profile_rcg_call
from boda_amd.rtc import make_rtc

net = sys.argv[1] if len(sys.argv) > 1 else "resnet-50"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 12
TILES = [("" if t == "auto" else t) for t in os.environ.get("TILES", "").split()] or [
    "", "128x128x64x2x2x2x1x32x2", "128x128x64x2x2x1x1x32x3", "128x128x32x2x2x2x1x32x3", "128x128x32x2x2x2x1x32x4", "64x128x64x1x4x2x1x32x3", "64x128x32x1x4x2x1x32x4",
    "128x64x64x2x2x2x1x32x3", "64x64x64x2x2x2x1x32x3", "64x64x32x2x2x2x1x32x4", "32x128x64x1x4x2x1x32x3", "256x128x32x4x2x1x1x32x3", "128x256x32x2x4x1x1x32x3", "128x256x64x2x4x1x1x32x2"]
rtc = make_rtc("(be=hip)", 0); rtc.init(); be = OpsBackend(rtc)
seen = {}
_b = int(os.environ.get("BATCH", "64"))
for op in (bench.alexnet_b256_ops(_b) if net == "alexnet" else bench.nin_ops(_b) if net == "nin" else bench.net_conv_ops(net, _b)):
    seen.setdefault(op.to_str(), op)
ops = list(seen.values())
sel = os.environ.get("SEL")
if sel:
    ops = [ops[int(i)] for i in sel.split(",")]
print("tiles:", TILES)
for i, op in enumerate(ops):
    g = op.conv_geom(); row = []; cfg = ""
    for t in TILES:
        try:
            anno = add_codegen_annotations(op, OpTune(hip_dtype="bf16", hip_layout="nhwc", hip_tile=t))
            outs, prc = profile_rcg_call(be, anno, 5, 0.0, iters, want_outs=False, tile=t)
            us = float(np.median(prc.all_secs[2:])) * 1e6
            row.append(f"{us:7.1f}" + ("*" if not t else " "))
            if not t:
                cfg = prc.launch["cfg"]
        except (UnsupErr, RtErr) as e:
            row.append("    n/a ")
    print(f"{i:2d} C{g['C']:4d} {g['H']:3d}x{g['W']:3d} OC{g['OC']:4d} k{g['KH']}s{g['SY']} {op.flops()/1e9:7.2f}GF [{cfg:22s}] " + " ".join(row), flush=True)
rtc.close()
