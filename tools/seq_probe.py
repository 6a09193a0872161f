#!/usr/bin/env python3
"""Per-launch time of AlexNet conv layers when the same layer repeats vs when layers alternate (in-sequence vs steady-state gap)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
if os.environ.get("WITH_TORCH"):
    import torch; torch.cuda.set_device(0); torch.cuda.synchronize(); print("torch initialised", flush=True)
import bench
from boda_amd.cnn_op import NATIVE_ARGS, OpTune, add_codegen_annotations
from boda_amd.rtc import make_rtc, RtcArg, RtcFuncCall, RtcFuncInfo
from boda_amd import gen_data as gd
rtc = make_rtc(); rtc.init(); rtc.compile(gd.func_infos())
ops = bench.alexnet_b256_ops(256)
calls = []
TILES = dict((int(kv.split(":")[0]), kv.split(":")[1]) for kv in os.environ.get("TILES", "").split(",") if kv)   # op index -> hip_tile
for i, op in enumerate(ops):
    anno = add_codegen_annotations(op, OpTune(hip_tile=TILES.get(i, ""), hip_dtype=os.environ.get("DTYPE", ""))); fn = anno.get_func_name(); g = f"{fn}__{i}"
    rtc.compile([RtcFuncInfo(g, "", [x for x, _ in NATIVE_ARGS[fn]], anno)])
    am = {}
    for an, io in NATIVE_ARGS[fn]:
        if io == "REF": am[an] = RtcArg.ref(anno.get_dims(an)); continue
        vn = f"{an}_{i}"; rtc.create_var_with_dims(vn, anno.get_dims(an)); am[an] = RtcArg.var(vn)
        if io == "IN": rtc.run(gd.gen_call("Convolution", an, vn, anno.get_dims(an), 5, 0.0))
    calls.append(RtcFuncCall(g, am))
rtc.finish_and_sync(); rtc.release_per_call_id_data()
def run1(i):
    if TILES: rtc.set_tune("conv_tile", TILES.get(i, ""))   # the tile tune is backend-global: set it per call
    return rtc.run(calls[i])
def run(seq, reps):
    for _ in range(3):
        for i in seq: run1(i)
    rtc.finish_and_sync(); rtc.release_per_call_id_data()
    ids = [[run1(i) for i in seq] for _ in range(reps)]
    rtc.finish_and_sync()
    t = np.array([[rtc.get_dur(c, c) for c in row] for row in ids]); wall = rtc.get_dur(ids[0][0], ids[-1][-1])
    rtc.release_per_call_id_data()
    return t.mean(0), wall / reps
SEQS = ([1] * 8, [2] * 8, [3] * 8, [1, 2, 3, 4], [1, 2, 3, 4, 5, 6, 7], [0, 1, 2, 3, 4, 5, 6, 7], [1, 1, 2, 2, 3, 3, 4, 4])
if os.environ.get("SEQ"):
    SEQS = ([int(x) for x in os.environ["SEQ"].split(",")],)
for seq in SEQS:
    t, wall = run(seq, 10)
    fl = [ops[i].flops() for i in seq]
    print("seq", seq, " TF/s per call:", " ".join(f"{f/ms/1e9:.0f}" for f, ms in zip(fl, t)), f"| wall/pass {wall:.3f} ms, sum kernels {t.sum():.3f} ms", flush=True)
if os.environ.get("SEQ"): sys.exit(0)
# clock behaviour under continuous load: conv2 repeated for ~2.5 s, mean TF/s per 100-launch window
import time
seq = [1]; fl = ops[1].flops()
out = []
t0 = time.perf_counter()
for w in range(14):
    ids = [rtc.run(calls[1]) for _ in range(100)]
    rtc.finish_and_sync()
    ms = np.array([rtc.get_dur(c, c) for c in ids]); rtc.release_per_call_id_data()
    out.append((time.perf_counter() - t0, fl / ms.mean() / 1e9, fl / ms.min() / 1e9))
print("continuous conv2: (t_s, mean TF/s, best TF/s) per 100 launches:", " ".join(f"({t:.2f},{a:.0f},{b:.0f})" for t, a, b in out), flush=True)
time.sleep(1.0)
ids = [rtc.run(calls[1]) for _ in range(30)]; rtc.finish_and_sync()
ms = np.array([rtc.get_dur(c, c) for c in ids]); rtc.release_per_call_id_data()
print("after 1 s idle: first 30 launches TF/s:", " ".join(f"{fl/m/1e9:.0f}" for m in ms), flush=True)
