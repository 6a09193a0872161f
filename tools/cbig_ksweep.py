#!/usr/bin/env python3
"""conv_big_f32.hip: where a launch's time goes -- t = a + b * K fits on one geometry (the fixed cost a = launch + prologue + epilogue, b = the K loop's rate), and the
kernel on an sgemm-shaped 1x1 convolution beside hip_sgemm.   usage: cbig_ksweep.py [--tiles t1,t2]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from boda_amd.cnn_op import OpTune, add_codegen_annotations
from boda_amd.op import parse_op
from boda_amd.ops_prof import OpsBackend, profile_rcg_call
from boda_amd.rtc import make_rtc
from tools.cbig_probe import conv_op

ap = argparse.ArgumentParser()
ap.add_argument("--tiles", default="128x512x16x2x4x1x1x32x2x2,256x256x16x2x4x1x1x32x2x2,128x128x16x2x4x2x1x32x2x2")
ap.add_argument("--iters", type=int, default=8)
a = ap.parse_args()
rtc = make_rtc(); rtc.init(); be = OpsBackend(rtc)

def t_of(op, tile):
    anno = add_codegen_annotations(op, OpTune(hip_tile=tile))
    _, prc = profile_rcg_call(be, anno, 5, run_iter=a.iters, want_outs=False, tile=tile)
    return min(prc.all_secs[1:]), prc.launch

for tile in [""] + a.tiles.split(","):
    pts = []
    for C in (64, 128, 256, 384, 512, 768):
        op = conv_op(256, C, 13, 13, 384, 3, 3, 1, 1)
        try:
            t, l = t_of(op, tile)
        except Exception as e:
            print(tile, C, "ERR", str(e)[:80]); continue
        pts.append((C * 9, t * 1e6))
        print(f"3x3 13x13 OC384 B256 C{C:4d} K{C*9:5d} tile {tile or 'auto':>28s} [{l['cfg']}] grid {l['grid']} {t*1e6:8.1f} us {op.flops()/t/1e12:6.1f} TF/s", flush=True)
    if len(pts) >= 3:
        K = np.array([p[0] for p in pts], float); T = np.array([p[1] for p in pts], float)
        b, a0 = np.polyfit(K[1:], T[1:], 1)
        fl_per_k = 2.0 * 384 * 256 * 169
        print(f"   fit over K >= {int(K[1])}: t = {a0:6.1f} us + {b*1e3:7.3f} ns * K  -> K-loop rate {fl_per_k/b/1e6:6.1f} TF/s, fixed cost {a0:5.1f} us", flush=True)

# sgemm-shaped 1x1: OC 4096, C 4096, 4096 pels (16 images of 16 x 16)
op = conv_op(16, 4096, 16, 16, 4096, 1, 1, 1, 0)
for tile in ["", "256x256x16x2x4x1x1x32x2x2", "256x256x16x2x4x1x1x32x4x2", "256x128x16x4x2x1x1x32x2x2"]:
    try:
        t, l = t_of(op, tile)
        print(f"1x1 4096^3-shaped tile {tile or 'auto':>28s} [{l['cfg']}] {l['kernel']} {t*1e6:8.1f} us {op.flops()/t/1e12:6.1f} TF/s", flush=True)
    except Exception as e:
        print(tile, "ERR", str(e)[:100])
sg = parse_op("(str_vals=(type=sgemm),nda_vals=(a=(dims=(K=4096,M=4096)),b=(dims=(K=4096,N=4096)),c=(dims=(M=4096,N=4096))))")
t, l = t_of(sg, "")
print(f"hip_sgemm 4096^3 [{l['cfg']}] {l['kernel']} {t*1e6:8.1f} us {sg.flops()/t/1e12:6.1f} TF/s")
