#!/usr/bin/env python3
"""Random-shape parity sweep: hip_conv (default plan: whatever operand mode / tile the planner picks) vs the CPU oracle, bit-exact.
usage: fuzz_conv.py [n_cases] [seed] [big|small] [hip_tile]    (GPU box; prints the failing shapes, exit code 1 on any mismatch)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np


def cases(n, seed, big=False):
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n:
        k = int(rng.choice([1, 1, 2, 3, 3, 3, 4, 5, 5, 7, 11]))
        kh, kw = (k, k) if rng.random() < 0.85 else (k, int(rng.choice([1, 2, 3, 5])))
        s = int(rng.choice([1, 1, 1, 2, 2, 3, 4]))
        p = int(rng.integers(0, max(kh, kw) // 2 + 2))
        h = int(rng.integers(max(kh - 2 * p, 1), 40)); w = int(rng.integers(max(kw - 2 * p, 1), 40))
        if h + 2 * p < kh or w + 2 * p < kw:
            continue
        if big:   # enough tiles for the 128x128 / 64x256 / 96x256 / 128x256 workgroup shapes
            b = int(rng.choice([16, 33, 64])); h = int(rng.integers(max(kh - 2 * p, 6), 60)); w = int(rng.integers(max(kw - 2 * p, 6), 60))
        else:
            b = int(rng.choice([1, 2, 3, 5, 8, 17]))
        c = int(rng.choice([1, 2, 3, 4, 7, 8, 16, 19, 32, 48, 64])); oc = int(rng.choice([1, 3, 8, 16, 24, 33, 64, 96, 100, 128, 160]))
        if 2.0 * b * ((h + 2 * p - kh) // s + 1) * ((w + 2 * p - kw) // s + 1) * oc * c * kh * kw > (4e10 if big else 3e9):
            continue
        out.append((b, c, h, w, oc, kh, kw, s, p))
    return out


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    big = len(sys.argv) > 3 and sys.argv[3] == "big"
    tile = sys.argv[4] if len(sys.argv) > 4 else ""      # optional explicit workgroup tile (hip_tile) for every case
    from boda_amd.cnn_op import OpTune
    from test_gpu_parity import _conv_op, _run
    from boda_amd.rtc import make_rtc
    from boda_amd.ops_prof import OpsBackend
    from oracle import boda_oracle as bo
    rtc = make_rtc("(be=hip)", 0); rtc.init(); be = OpsBackend(rtc)
    bad = 0; modes = {}
    for sh in cases(n, seed, big):
        op = _conv_op(*sh)
        outs, prc = _run(be, op, 5, tune=OpTune(hip_tile=tile), include_ins=True)
        g = op.conv_geom()
        want = bo.conv_fwd(outs["in"], outs["filts"], outs["biases"], (g["SY"], g["SX"]), (g["PY"], g["PX"]), True)
        cfg = prc.launch["cfg"]; modes[cfg] = modes.get(cfg, 0) + 1
        if not np.array_equal(want, outs["out"]):
            bad += 1; print("MISMATCH", sh, cfg, int((want != outs["out"]).sum()), "of", want.size, flush=True)
    print(f"{n} cases, {bad} mismatches; tile configs used:", dict(sorted(modes.items(), key=lambda kv: -kv[1])))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
