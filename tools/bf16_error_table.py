"""GPU tool: per-layer error table of the channels-last bf16 convolutions at the BENCH batch (config 5: GoogLeNet-conv and ResNet-50 at 64 images).

  python tools/bf16_error_table.py [batch] > profiles/r05_bf16_error_table.txt

For every distinct conv of the two nets: the kernel runs at `batch` images (float and bf16 outputs, the plan the bench takes -- K slices included); the CPU oracle is fed
the same bf16-rounded operands for the FIRST TWO images (outputs of an image do not depend on the batch).  Columns:
  K        contraction length (in_chan x ky x kx)
  mrd      max |got - want| / max(1, |got|, |want|), float output           (the reference's compare: src/boda_base.cc:140-154)
  /emp     mrd / the empirical per-layer bound of the tests  1e-3 max(1, sqrt(K / 2400))
  /derived max over outputs of |got - want| / (2 (K + 1) 2^-24 sum_k |in_k||filts_k|)   -- the bound that follows from the arithmetic (any summation order); < 1 required
  bf16 /d  the same ratio for the bf16 output with its extra 2^-8 (|want| + that bound): one round-to-nearest-even, half a bf16 ulp <= 2^-8 of the fp32 result
           (a value just under 1 is the rounding itself, not the summation)
Parity for bf16 is UNPINNED by construction (the reference has no bf16): these are measured distances from a restated fp32 oracle on rounded operands.
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from boda_amd.cnn_op import OpTune, add_codegen_annotations
from boda_amd.ops_prof import OpsBackend, profile_rcg_call
from boda_amd.rtc import make_rtc
from oracle import boda_oracle as bo

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rtc = make_rtc("(be=hip)", 0); rtc.init(); be = OpsBackend(rtc)
print(__doc__)
worst = {}
for net in ("googlenet_conv", "resnet-50"):
    seen = {}
    for op in bench.net_conv_ops(net, batch):
        seen.setdefault(op.to_str(), op)
    print(f"## {net}, {batch} images ({len(seen)} distinct convolutions)")
    print(f"{'layer':34s} {'K':>6s} {'plan':26s} {'mrd':>9s} {'/emp':>6s} {'/derived':>9s} {'bf16 /d':>8s}")
    for op in seen.values():
        g = op.conv_geom(); K = g["C"] * g["KH"] * g["KW"]
        row = {}
        for out in ("f32", ""):
            anno = add_codegen_annotations(op, OpTune(hip_dtype="bf16", hip_layout="nhwc", hip_out=out))
            outs, prc = profile_rcg_call(be, anno, 5, 0.0, 1, include_ins=True)
            xin = bo.to_bf16(outs["in"][:2]); f = bo.to_bf16(outs["filts"]); b = outs["biases"]
            want = bo.conv_fwd(xin, f, b, (g["SY"], g["SX"]), (g["PY"], g["PX"]), True).astype(np.float64)
            S = bo.conv_fwd(np.abs(xin), np.abs(f), np.abs(b), (g["SY"], g["SX"]), (g["PY"], g["PX"]), False).astype(np.float64)
            got = outs["out"][:2].astype(np.float64)
            err = np.abs(got - want)
            d = 2.0 * (K + 1) * 2.0 ** -24 * S
            lim = d + (0.0 if out == "f32" else 2.0 ** -8 * (np.abs(want) + d))
            row[out] = (float((err / np.maximum(1.0, np.maximum(np.abs(got), np.abs(want)))).max()), float((err / np.maximum(lim, 1e-300)).max()), prc.launch["cfg"])
        emp = 1e-3 * max(1.0, (K / 2400.0) ** 0.5)
        name = f"C{g['C']} {g['H']}x{g['W']} OC{g['OC']} k{g['KH']}s{g['SY']}p{g['PY']}"
        print(f"{name:34s} {K:6d} {row['f32'][2]:26s} {row['f32'][0]:9.2e} {row['f32'][0]/emp:6.3f} {row['f32'][1]:9.4f} {row[''][1]:8.4f}", flush=True)
        for k, v in (("mrd/emp", row["f32"][0] / emp), ("f32/derived", row["f32"][1]), ("bf16/derived", row[""][1])):
            worst[k] = max(worst.get(k, 0.0), v)
print("## worst over all layers:", ", ".join(f"{k} {v:.4f}" for k, v in worst.items()))
rtc.close()
