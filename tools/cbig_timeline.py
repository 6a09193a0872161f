#!/usr/bin/env python3
"""Timeline of one conv_big_f32.hip launch from in-kernel clock stamps (-DTSTAMP=1, BODAHIP_CBIG_TSTAMP): when workgroups start, how long the prologue, the K loop, the
store issue and the store drain take, and when the last one ends.   usage: cbig_timeline.py --spec alexnet:256 --op 2 --tile 128x512x16x2x4x1x1x32x2x2"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--spec", default="alexnet:256"); ap.add_argument("--op", type=int, default=2); ap.add_argument("--tile", default="128x512x16x2x4x1x1x32x2x2")
a = ap.parse_args()
F = "/tmp/cbig_ts.txt"
if os.path.exists(F): os.remove(F)
os.environ["BODAHIP_CBIG_TSTAMP"] = F
os.environ["BODAHIP_EXTRA_DEFS"] = (os.environ.get("BODAHIP_EXTRA_DEFS", "") + " -DTSTAMP=1").strip()
import numpy as np
from boda_amd.cnn_op import OpTune, add_codegen_annotations
from boda_amd.ops_prof import OpsBackend, profile_rcg_call
from boda_amd.rtc import make_rtc
import bench
wl, batch = a.spec.split(":")
ops = {"alexnet": lambda: bench.alexnet_b256_ops(int(batch)), "nin": lambda: bench.nin_ops(int(batch))}[wl]()
op = ops[a.op]
rtc = make_rtc(); rtc.init(); be = OpsBackend(rtc)
anno = add_codegen_annotations(op, OpTune(hip_tile=a.tile))
_, prc = profile_rcg_call(be, anno, 5, run_iter=5, want_outs=False, tile=a.tile)
print("event times (us):", " ".join(f"{t*1e6:.1f}" for t in prc.all_secs), prc.launch["cfg"])
launches = []; cur = None
for line in open(F):
    if line.startswith("launch"): cur = []; launches.append(cur)
    else: cur.append([int(x) for x in line.split()])
for li, L in enumerate(launches[-2:]):
    T = np.array(L, dtype=np.float64); t0 = T[:, 0].min()
    us = lambda x: x / 100.0   # 100 MHz
    st, pro, loop, iss, drain = us(T[:, 0] - t0), us(T[:, 2] - T[:, 0]), us(T[:, 3] - T[:, 2]), us(T[:, 4] - T[:, 3]), us(T[:, 5] - T[:, 4])
    end = us(T[:, 5] - t0); cyc = T[:, 7] - T[:, 6]
    q = lambda v: f"min {v.min():7.1f} med {np.median(v):7.1f} max {v.max():7.1f}"
    print(f"launch {li}: {len(L)} workgroups, last stamp at {end.max():.1f} us")
    print("  start    ", q(st)); print("  prologue ", q(pro)); print("  K loop   ", q(loop)); print("  store iss", q(iss)); print("  drain    ", q(drain)); print("  end      ", q(end)); print(f"  K loop shader cycles med {np.median(cyc):.0f} -> clock {np.median(cyc / np.maximum(loop, 1e-9)) / 1e3:.3f} GHz")
    bw = np.array([int(r[1]) >> 4 for r in L], dtype=np.float64)
    print(f"  wave 0 at the K loop's barriers: med {np.median(bw):.0f} cycles = {100 * np.median(bw / np.maximum(cyc, 1)):.1f} % of the loop")
    if T.shape[1] >= 12: print("  first staging wave, shader cycles (med): filter part %.0f | pel part %.0f | LDS drain %.0f | at the barrier %.0f" % tuple(np.median(T[:, e]) for e in (8, 9, 10, 11)))
    xcc = np.array([int(r[1]) & 15 for r in L])
    for x in sorted(set(xcc)): m = xcc == x; print(f"  xcc {x}: n {m.sum():3d} loop med {np.median(loop[m]):7.1f} end max {end[m].max():7.1f}")
