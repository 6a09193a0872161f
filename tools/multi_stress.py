#!/usr/bin/env python3
"""Stress the multi-problem launch: N repetitions per tile, each compared bit for bit with the members' separate launches (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_gpu_nhwc as T
from boda_amd import nhwc
from boda_amd.cnn_op import OpTune
from boda_amd.ops_prof import OpsBackend
from boda_amd.rtc import make_rtc, RtcArg, RtcFuncCall, RtcFuncInfo
rtc = make_rtc(); rtc.init(); be = OpsBackend(rtc)
N = int(os.environ.get("REPS", "40"))
tune = OpTune(hip_dtype="bf16", hip_layout="nhwc", hip_patch=0, hip_s2d=0, hip_tile="64x64x64x2x2x2x1")
ol = T._OpList(rtc, T.MULTI_SHAPES, lambda i: tune, "st")
want = []
for i, c in enumerate(ol.calls):
    rtc.run(c); want.append(ol.out(i))
# the separate launches against themselves first
bad = 0
for rep in range(10):
    ol.zero_outs()
    for c in ol.calls: rtc.run(c)
    rtc.finish_and_sync()
    bad += sum(not np.array_equal(ol.out(i), want[i]) for i in range(len(ol.calls)))
print("separate launches vs themselves: mismatching members over 10 reps:", bad, flush=True)
for ti, tile in enumerate(os.environ.get("TILES", ",32x128x64x1x4x2x1x32x2,64x128x64x2x2x3x1x32x2,64x128x64x2x2x2x1x32x3,64x64x32x2x2x2x1x32x3,128x64x32x2x2x2x1x32x4").split(",")):
    manno = nhwc.annotate_multi(ol.annos)
    if tile: manno.str_vals["hip_tile"] = tile
    else: manno.str_vals.pop("hip_tile", None)
    fn = f"st_multi_{ti}"
    rtc.compile([RtcFuncInfo(fn, "", nhwc.multi_arg_names(len(ol.calls)), manno)])
    am = {"multi": RtcArg.ref(manno.get_dims("multi"))}
    for m, c in enumerate(ol.calls):
        for an in ("filts", "biases", "in", "stride", "in_pad", "out"): am[f"{an}_{m}"] = c.arg_map[an]
    fails = {}
    for rep in range(N):
        ol.zero_outs(); rtc.run(RtcFuncCall(fn, am)); rtc.finish_and_sync()
        for i in range(len(ol.calls)):
            g = ol.out(i)
            if not np.array_equal(g, want[i]):
                fails.setdefault(i, []).append(int((g != want[i]).sum()))
    rtc.release_per_call_id_data()
    print(f"tile {tile or '(default)':28s} {rtc.last_launch()['cfg']:22s} reps {N} failing members:", {T.MULTI_SHAPES[i][:5]: v for i, v in fails.items()}, flush=True)
ol.release()
