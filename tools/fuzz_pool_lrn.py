#!/usr/bin/env python3
"""Random pool / LRN geometries through ConvPipeFwd: the geometry-specialised kernels (channels-last bf16 and the fp32 template's unconditional-taps form) against the
generic ones and the oracle.   usage: fuzz_pool_lrn.py [n_pipes] [seed]   (GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    from boda_amd.cnn_op import OpTune
    from boda_amd.conv_pipe import ConvPipe, ConvPipeFwd, PipeOp
    from boda_amd.op import Dims, RtErr, UnsupErr
    from boda_amd.rtc import make_rtc
    from oracle import boda_oracle as bo
    rng = np.random.default_rng(seed)
    rtc = make_rtc("(be=hip)", 0); rtc.init()
    bad = 0; n_ops = 0
    for it in range(n):
        B = int(rng.choice([1, 2, 5])); C = int(rng.choice([8, 24, 40, 64, 16, 32])); H = int(rng.integers(3, 30)); W = int(rng.integers(3, 30))
        def pipe():
            r = np.random.default_rng(seed * 1000 + it)
            p = ConvPipe("fz", "data", Dims.make("float", img=B, chan=C, y=H, x=W)); k = 0
            for _ in range(10):
                kh = int(r.choice([1, 2, 3, 3, 4, 5, 7])); kw = kh if r.random() < 0.8 else int(r.choice([1, 2, 3, 5]))
                s = int(r.choice([1, 1, 2, 2, 3])); pd = int(r.integers(0, max(1, min(kh, kw))))
                if kh > H + 2 * pd or kw > W + 2 * pd: continue
                try:
                    p.add(PipeOp(f"p{k}", "Pooling", "data", f"p{k}", kern_sz=(kh, kw), stride=(s, s), in_pad=(pd, pd), avg_pool=int(r.random() < 0.4))); k += 1
                except RtErr:
                    pass
            for j, ls in enumerate((3, 5, 7)):
                p.add(PipeOp(f"l{j}", "LRN", "data", f"l{j}", lrn=(ls, float(r.choice([1e-4, 2e-2, 0.5])), 0.75, float(r.choice([1.0, 2.0])))))
            return p
        data = (bo.gen_conv_in(B, C, H, W) * np.float32(3.0)).astype(np.float32)
        for mode in ("nhwc", "f32"):
            res = {}
            for spec in (True, False):
                cp = pipe()
                fwd = ConvPipeFwd(rtc, OpTune(hip_dtype="bf16", hip_layout="nhwc") if mode == "nhwc" else None, spec_fwd=spec)
                try:
                    fwd.init(cp, op_params={})
                    io = {"data": data}; fwd.run_fwd(["data"], io, [o.top for o in cp.ops]); res[spec] = io
                finally:
                    fwd.release()
            x = bo.to_bf16(data) if mode == "nhwc" else data
            ulp = 2.0 ** -8 if mode == "nhwc" else 2.0 ** -22
            for op in pipe().ops:
                n_ops += 1
                g, gen = res[True][op.top].astype(np.float64), res[False][op.top].astype(np.float64)
                if op.type == "Pooling":
                    w = bo.pool_fwd(x, op.kern_sz, op.stride, op.in_pad, bool(op.avg_pool)).astype(np.float64)
                    # (a window that lies wholly in the padding -- stride > window, which the reference's output-size rule admits -- averages to 0/0: NaN everywhere alike)
                    nn = ~np.isnan(w) & (np.abs(w) < 3e38)      # (and its maximum is -FLT_MAX, which a bf16 tensor stores as -inf)
                    # max: exact three ways.  average: the device divides through v_rcp_f32 (fast-math build) -- the specialised and the generic kernel may round the
                    # quotient one ulp apart (seen on non-square windows), either is within one rounding of the oracle's IEEE division
                    same = np.array_equal(g, gen, equal_nan=True) if not op.avg_pool else bool((np.abs(g[nn] - gen[nn]) <= 2 * ulp * np.abs(gen[nn]) + 1e-6).all())
                    ok = g.shape == w.shape and same and np.array_equal(np.isnan(g), np.isnan(w)) and \
                        (np.array_equal(w[nn], g[nn]) if not op.avg_pool else bool((np.abs(g[nn] - w[nn]) <= 2 * ulp * np.abs(w[nn]) + 1e-6).all()))
                else:
                    w = bo.lrn_fwd(x, *op.lrn).astype(np.float64)
                    # (the carried sum of squares loses digits when alpha is large -- the draw includes 0.5 -- in the oracle's order as in the kernels': 5e-5 relative covers it)
                    ok = bool((np.abs(g - w) <= 2 * max(ulp, 2.5e-5) * np.abs(w) + 1e-5).all()) and bool((np.abs(g - gen) <= 2 * max(ulp, 2.5e-5) * np.abs(gen) + 1e-6).all())
                if not ok:
                    bad += 1; print("MISMATCH", mode, (B, C, H, W), op.tag, op.type, op.kern_sz, op.stride, op.in_pad, op.avg_pool, op.lrn, float(np.abs(g - w).max()), float(np.abs(g - gen).max()), flush=True)
    print(f"{n} pipes, {n_ops} ops, {bad} mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
