#!/usr/bin/env python3
"""Tile-starved long-K fp32 layers under deeper register-ring prefetch (PF = K-tiles in flight, gemm_conv_f32.hip): time and bit-equality with the planner's choice.
   usage (GPU box): fc_pf_sweep.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from boda_amd.cnn_op import OpTune, add_codegen_annotations
from boda_amd.ops_prof import OpsBackend, profile_rcg_call
from boda_amd.rtc import make_rtc
from test_gpu_parity import _conv_op, _sgemm_op
rtc = make_rtc(); rtc.init(); be = OpsBackend(rtc)
LAYERS = [("fc6", _conv_op(256, 256, 6, 6, 4096, 6, 6, 1, 0)), ("fc7", _conv_op(256, 4096, 1, 1, 4096, 1, 1, 1, 0)), ("fc8", _conv_op(256, 4096, 1, 1, 1000, 1, 1, 1, 0)),
          ("fc6@128", _conv_op(128, 256, 6, 6, 4096, 6, 6, 1, 0)), ("g-cls", _conv_op(64, 1024, 1, 1, 1000, 1, 1, 1, 0)),
          ("sgemm2048", _sgemm_op(2048, 2048, 2048)), ("sgemm1024", _sgemm_op(1024, 1024, 1024)), ("sgemm3072", _sgemm_op(3072, 3072, 3072))]
TILES = ["", "64x64x32x2x2x1x1x32x2", "64x64x32x2x2x1x1x32x4", "64x64x32x2x2x1x1x32x6", "64x64x32x2x2x1x1x32x8", "64x64x16x2x2x1x1x32x8", "64x64x64x2x2x1x1x32x4",
         "32x32x64x2x2x1x1x16x2", "32x32x64x2x2x1x1x16x4", "32x32x64x2x2x1x1x16x8", "32x32x32x2x2x1x1x16x8", "32x64x32x2x4x1x1x16x4", "32x64x32x2x4x1x1x16x8", "32x64x64x2x4x1x1x16x4",
         "64x32x32x4x2x1x1x16x4", "64x64x32x4x4x1x1x16x4", "64x64x32x4x4x1x1x16x8", "128x128x16x2x2x1x1x32x4", "128x64x32x2x2x1x1x32x4"]
for nm, op in LAYERS:
    ref = None
    for t in TILES:
        try:
            anno = add_codegen_annotations(op, OpTune(hip_tile=t))
            outs, prc = profile_rcg_call(be, anno, 5, 0.0, 14, tile=t)
            ms = sorted(prc.all_secs[3:])[len(prc.all_secs[3:]) // 4] * 1e3
            o = outs["out" if "out" in outs else "c"]
            if ref is None: ref = o
            print(f"{nm:10s} tile {t or '(auto)':26s} {prc.launch['cfg']:26s} {ms*1e3:8.1f} us {op.flops()/ms/1e9:7.1f} TF/s  {'same' if np.array_equal(ref, o) else 'DIFF'}", flush=True)
        except Exception as e:
            print(f"{nm:10s} tile {t:26s} {type(e).__name__}: {str(e)[:100]}", flush=True)
