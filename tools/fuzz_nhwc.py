#!/usr/bin/env python3
"""Random-shape parity sweep of hip_conv_nhwc (channels-last bf16: implicit GEMM and the input-patch kernel, planner tiles and random forced tiles, float and
bf16 outputs) against the oracle on bf16-rounded operands, with the bounds of tests/test_gpu_nhwc.py.   usage: fuzz_nhwc.py [n_cases] [seed]   (GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    from boda_amd.cnn_op import OpTune
    from boda_amd.op import UnsupErr
    import test_gpu_nhwc as T
    from boda_amd.rtc import make_rtc
    from boda_amd.ops_prof import OpsBackend
    rng = np.random.default_rng(seed)
    rtc = make_rtc("(be=hip)", 0); rtc.init(); be = OpsBackend(rtc)
    tiles = ["", "", "", "128x128x0x4x1", "64x256x0x2x2", "64x128x0x2x2", "128x64x0x4x1", "32x128x0x1x4", "256x128x0x4x1x1", "64x64x0x2x2", "128x256x0x4x2x1"]
    bad = 0; used = {}; done = 0
    while done < n:
        k = int(rng.choice([2, 3, 3, 3, 4, 5, 5, 7])); kh, kw = (k, k) if rng.random() < 0.8 else (k, int(rng.choice([1, 2, 3, 5])))
        if kh * kw < 2: continue
        p = int(rng.integers(0, max(kh, kw) // 2 + 2)); s = 1 if rng.random() < 0.85 else 2
        b = int(rng.choice([1, 2, 3, 5, 9, 17, 40])); h = int(rng.integers(max(kh - 2 * p, 1), 34)); w = int(rng.integers(max(kw - 2 * p, 1), 34))
        c = int(rng.choice([3, 8, 16, 24, 32, 40, 56, 64, 72, 96, 136])); oc = int(rng.choice([1, 8, 16, 24, 33, 48, 64, 96, 100, 128, 160, 208]))
        if h + 2 * p < kh or w + 2 * p < kw: continue
        if 2.0 * b * ((h + 2 * p - kh) // s + 1) * ((w + 2 * p - kw) // s + 1) * oc * c * kh * kw > 3e9: continue
        sh = (b, c, h, w, oc, kh, kw, s, p); op = T._conv_op(*sh)
        tile = tiles[int(rng.integers(0, len(tiles)))]; f32 = rng.random() < 0.5
        try:
            outs, prc = T._run(be, op, OpTune(hip_tile=tile, **(T.NHWC_F32 if f32 else T.NHWC)))
        except UnsupErr as e:
            if tile: continue          # (a forced tile the layer's kernel does not take)
            bad += 1; done += 1; print("UNSUPPORTED", sh, str(e)[:200], flush=True); continue
        done += 1
        key = prc.launch["kernel"].replace("bodahip_", "") + " " + prc.launch["cfg"]; used[key] = used.get(key, 0) + 1
        try:
            (T._check_f32 if f32 else T._check_bf16)(op, outs, prc)
        except AssertionError as e:
            bad += 1; print("MISMATCH", sh, tile, "f32" if f32 else "bf16", str(e)[:300], flush=True)
    print(f"{n} cases, {bad} mismatches; kernels / tiles:", dict(sorted(used.items(), key=lambda kv: -kv[1])))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
