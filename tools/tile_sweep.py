#!/usr/bin/env python3
"""Time one op (or a named workload) under several native-kernel tiles.  usage:
   tile_sweep.py --workload alexnet|sgemm --ops 1,2 --tiles 128x128x16x2x2x2,128x128x32x2x2x2 [--iters 5]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from boda_amd.cnn_op import OpTune, add_codegen_annotations
from boda_amd.ops_prof import OpsBackend, profile_rcg_call
from boda_amd.rtc import make_rtc
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="alexnet")
ap.add_argument("--ops", default="")
ap.add_argument("--tiles", default="")
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--dtype", default="")
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--mode", type=int, default=5)   # gen_data mode: 5 = the reference's random pattern, 1 = all zeros (clock / power experiments)
a = ap.parse_args()
ops = {"alexnet": lambda: bench.alexnet_b256_ops(a.batch), "nin": lambda: bench.nin_ops(a.batch), "sgemm": bench.sgemm_full_ops,
       "googlenet": lambda: bench.net_conv_ops("googlenet_conv", a.batch), "resnet50": lambda: bench.net_conv_ops("resnet-50", a.batch)}[a.workload]()
sel = [int(x) for x in a.ops.split(",")] if a.ops else list(range(len(ops)))
tiles = [""] + [t for t in a.tiles.split(",") if t]
rtc = make_rtc(); rtc.init(); be = OpsBackend(rtc)
for i in sel:
    op = ops[i]
    for t in tiles:
        try:
            anno = add_codegen_annotations(op, OpTune(hip_tile=t, hip_dtype=a.dtype))
            _, prc = profile_rcg_call(be, anno, a.mode, run_iter=a.iters, want_outs=False, tile=t)
            best = min(prc.all_secs[1:]) if len(prc.all_secs) > 1 else prc.all_secs[0]
            g = op.conv_geom() if op.get_type() == "Convolution" else op.sgemm_geom()
            desc = (f"C{g['C']} {g['H']}x{g['W']} OC{g['OC']} k{g['KH']}s{g['SY']}" if "OC" in g else f"M{g['M']} N{g['N']} K{g['K']}")
            print(f"op {i:2d} {desc:28s} tile {t or 'auto':>24s} [{prc.launch['cfg']:>22s}] grid {prc.launch['grid']:6d}  {best*1e3:9.4f} ms  {op.flops()/best/1e12:7.2f} TF/s  {op.algo_bytes()/best/1e9:7.0f} GB/s", flush=True)
        except Exception as e:
            print(f"op {i:2d} tile {t:>24s} ERR {type(e).__name__}: {str(e)[:100]}", flush=True)
