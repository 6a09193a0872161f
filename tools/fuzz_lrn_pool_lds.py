#!/usr/bin/env python3
"""Random LRN -> max-pooling pairs through ConvPipeFwd's default mode (the pair as ONE workgroup kernel through LDS, boda_amd/nhwc.py LRN_POOL_LDS_SRC) against the two kernels
run apart, bit for bit, and the pairs' pooled outputs against the oracle's pooling of the device's LRN output.   usage: fuzz_lrn_pool_lds.py [n_pipes] [seed]   (GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from boda_amd.cnn_op import OpTune
from boda_amd.conv_pipe import ConvPipe, ConvPipeFwd, PipeOp
from boda_amd.op import Dims, RtErr
from boda_amd.rtc import make_rtc
from oracle import boda_oracle as bo

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
rtc = make_rtc("(be=hip)", 0); rtc.init()
bad = n_pairs = n_lds = 0
for it in range(n):
    B = int(rng.choice([1, 2, 3, 5])); C = int(rng.choice([8, 16, 24, 40, 64, 96, 192, 256])); H = int(rng.integers(3, 40)); W = int(rng.integers(3, 40))
    def pipe():
        r = np.random.default_rng(seed * 1000 + it)
        p = ConvPipe("fz", "data", Dims.make("float", img=B, chan=C, y=H, x=W)); k = 0
        for _ in range(8):
            kh = int(r.choice([2, 3, 3, 3, 4, 5])); kw = kh if r.random() < 0.8 else int(r.choice([1, 2, 3, 5]))
            s = int(r.choice([1, 2, 2, 2, 3])); pd = int(r.integers(0, max(1, min(kh, kw)))) if r.random() < 0.4 else 0
            if kh > H + 2 * pd or kw > W + 2 * pd: continue
            ls = int(r.choice([3, 5, 5, 7, 9]))
            try:
                p.add(PipeOp(f"l{k}", "LRN", "data", f"l{k}", lrn=(ls, float(r.choice([1e-4, 2e-2, 0.5])), float(r.choice([0.75, 0.5])), float(r.choice([1.0, 2.0])))))
                p.add(PipeOp(f"p{k}", "Pooling", f"l{k}", f"p{k}", kern_sz=(kh, kw), stride=(s, s), in_pad=(pd, pd))); k += 1
            except RtErr:
                pass
        return p
    data = (bo.gen_conv_in(B, C, H, W) * np.float32(3.0)).astype(np.float32)
    res = {}
    for mode in ("pool_first", False):
        cp = pipe()
        fwd = ConvPipeFwd(rtc, OpTune(hip_dtype="bf16", hip_layout="nhwc"), fuse_pool_lrn=mode)
        try:
            fwd.init(cp, op_params={})
            if mode: n_lds += len(fwd.lds_pool_lrn); n_pairs += sum(1 for o in cp.ops if o.type == "Pooling")
            io = {"data": data}; fwd.run_fwd(["data"], io, [o.top for o in cp.ops]); res[mode] = io
        finally:
            fwd.release()
    for op in pipe().ops:
        a, b = res["pool_first"][op.top], res[False][op.top]
        ok = np.array_equal(a, b, equal_nan=True)
        if ok and op.type == "Pooling":      # against the oracle: the pooling of the LRN output the device produced (bf16 values: the maximum is exact)
            w = bo.pool_fwd(res[False][op.bot], op.kern_sz, op.stride, op.in_pad, False)
            nn = np.abs(w) < 3e38
            ok = a.shape == w.shape and np.array_equal(a[nn], w[nn])
        if not ok:
            bad += 1; print(f"MISMATCH pipe {it} B{B} C{C} {H}x{W} {op.tag} k{op.kern_sz} s{op.stride} p{op.in_pad} lrn{getattr(op, 'lrn', None)}: {int((a != b).sum())} elements differ", flush=True)
print(f"{n} pipes, {n_pairs} LRN -> pooling pairs, {n_lds} through LDS, {bad} mismatches")
sys.exit(1 if bad else 0)
