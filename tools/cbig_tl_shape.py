#!/usr/bin/env python3
"""cbig_timeline on an arbitrary convolution shape: cbig_tl_shape.py B C H W OC KH S P tile [mode]   (mode: gen_data mode, 5 = random, 1 = zeros)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
F = "/tmp/cbig_ts.txt"
if os.path.exists(F): os.remove(F)
os.environ["BODAHIP_CBIG_TSTAMP"] = F
os.environ["BODAHIP_EXTRA_DEFS"] = (os.environ.get("BODAHIP_EXTRA_DEFS", "") + " -DTSTAMP=1").strip()
import numpy as np
from boda_amd.cnn_op import OpTune, add_codegen_annotations
from boda_amd.ops_prof import OpsBackend, profile_rcg_call
from boda_amd.rtc import make_rtc
from tools.cbig_probe import conv_op
B, C, H, W, OC, KH, S, P = [int(x) for x in sys.argv[1:9]]; tile = sys.argv[9]; mode = int(sys.argv[10]) if len(sys.argv) > 10 else 5
op = conv_op(B, C, H, W, OC, KH, KH, S, P)
rtc = make_rtc(); rtc.init(); be = OpsBackend(rtc)
anno = add_codegen_annotations(op, OpTune(hip_tile=tile))
_, prc = profile_rcg_call(be, anno, mode, run_iter=5, want_outs=False, tile=tile)
L = []
for line in open(F):
    if line.startswith("launch"): L = []
    else: L.append([int(x) for x in line.split()])
T = np.array(L, dtype=np.float64)
loop = (T[:, 3] - T[:, 2]) / 100.0; cyc = T[:, 7] - T[:, 6]
nkt = (C * KH * KH + int(prc.launch["cfg"].split("x")[2].split("_")[0]) - 1) // int(prc.launch["cfg"].split("x")[2].split("_")[0])
print(f"{sys.argv[1:9]} {prc.launch['cfg']} mode {mode}: K loop med {np.median(loop):.1f} us, {np.median(cyc):.0f} cycles, clock {np.median(cyc/loop)/1e3:.3f} GHz, flops/wg-loop -> {2.0*C*KH*KH*OC*B*((H+2*P-KH)//S+1)**2/ (np.median(loop)*1e-6) / 1e12 * (len(L) / max(1, len(L))):.1f} TF/s if all workgroups ran concurrently; staging wave at barrier {np.median(T[:,11]):.0f}")
