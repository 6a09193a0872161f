#!/usr/bin/env python3
"""Per-layer tile tuning (the op-tuner idea of the reference, src/op-tuner.cc, applied to the native kernels' one free parameter):
time every distinct conv of a workload under the planner's choice and under a list of candidate tiles (all of them bit-exact: no
split-K), write the winners to a tile-wisdom file   <canonical op line> TAB <tile> TAB <auto ms> TAB <best ms>
usage: tune_tiles.py --workload resnet50 --batch 64 --out tests/golden/tile_wisdom/resnet50_b64.txt"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import bench
from boda_amd.cnn_op import OpTune, add_codegen_annotations
from boda_amd.ops_prof import OpsBackend, profile_rcg_call
from boda_amd.rtc import make_rtc
CANDS = "64x64x32x2x2x2x1x32x2,128x128x16x2x2x2,64x256x16x1x4x2,128x256x16x2x4x1,32x64x16x2x4x2x1x16x2,96x128x16x1x4x2,64x128x16x1x4x2,128x64x16x2x2x2,32x256x16x1x4x2,32x128x16x1x4x2"
ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="resnet50"); ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--tiles", default=CANDS); ap.add_argument("--iters", type=int, default=12); ap.add_argument("--out", default="")
ap.add_argument("--min-gain", type=float, default=0.03, help="keep a tile only if it beats the planner's choice by this fraction")
a = ap.parse_args()
ops = {"alexnet": lambda: bench.alexnet_b256_ops(a.batch), "nin": lambda: bench.nin_ops(a.batch),
       "googlenet": lambda: bench.net_conv_ops("googlenet_conv", a.batch), "resnet50": lambda: bench.net_conv_ops("resnet-50", a.batch)}[a.workload]()
rtc = make_rtc(); rtc.init(); be = OpsBackend(rtc)
def t_of(op, tile):
    anno = add_codegen_annotations(op, OpTune(hip_tile=tile))
    _, prc = profile_rcg_call(be, anno, 5, run_iter=a.iters, want_outs=False, tile=tile)
    s = sorted(prc.all_secs[2:]); return s[len(s) // 4] * 1e3, prc.launch["cfg"]      # lower quartile of the warm launches, ms
seen = {}; tot_auto = tot_best = 0.0; lines = []
for op in ops:
    k = op.to_str()
    if k not in seen:
        auto_ms, auto_cfg = t_of(op, "")
        best_ms, best_t, best_cfg = 1e30, "", ""
        for t in a.tiles.split(","):
            if not t: continue
            try: ms, cfg = t_of(op, t)
            except Exception: continue
            if ms < best_ms: best_ms, best_t, best_cfg = ms, t, cfg
        # The planner's choice is timed first, right after the data was generated (clocks not settled): an identical configuration timed later came out 8-12 % "faster".
        # So: the planner's choice again at the end, and the winner and the planner's choice alternately twice more -- the minimum of each side counts.
        auto_ms = min(auto_ms, t_of(op, "")[0])
        if best_t and best_cfg != auto_cfg:
            for _ in range(2):
                best_ms = min(best_ms, t_of(op, best_t)[0]); auto_ms = min(auto_ms, t_of(op, "")[0])
        if (not best_t) or best_cfg == auto_cfg or best_ms > auto_ms * (1.0 - a.min_gain): best_ms, best_t = auto_ms, ""
        seen[k] = (auto_ms, best_ms, best_t, auto_cfg)
        g = op.conv_geom()
        print(f"C{g['C']:4d} {g['H']:3d}x{g['W']:<3d} OC{g['OC']:4d} k{g['KH']}s{g['SY']}  auto {auto_cfg:>22s} {auto_ms*1e3:7.1f} us   best {best_t or '(auto)':>26s} {best_ms*1e3:7.1f} us", flush=True)
        if best_t: lines.append(f"{k}\t{best_t}\t{auto_ms:.5f}\t{best_ms:.5f}")
    tot_auto += seen[k][0]; tot_best += seen[k][1]
print(f"{a.workload} B={a.batch}: sum of layer times auto {tot_auto:.3f} ms -> tuned {tot_best:.3f} ms ({100*(tot_auto/tot_best-1):.1f} % faster), {len(lines)} of {len(seen)} distinct layers re-tiled")
if a.out:
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    open(a.out, "w").write("".join(l + "\n" for l in lines))
