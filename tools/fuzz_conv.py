#!/usr/bin/env python3
"""Random-shape parity sweep: hip_conv (default plan: whatever operand mode / tile the planner picks) vs the CPU oracle, bit-exact.
usage: fuzz_conv.py [n_cases] [seed] [big|small] [hip_tile]    (GPU box; prints the failing shapes, exit code 1 on any mismatch)
MODE=exact (default) | bf16 (hip_conv_bf16 incl. the LDS-patch kernel; mrd < 1e-3 vs the oracle on bf16-rounded operands) |
     winograd (conv_algo=winograd_all; mrd < 2e-3, the reference's Winograd bound) | k1s (1x1 stride-1 shapes through random k1_stream
     specs; bit-exact)"""
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


def cases_mode(n, seed, big, mode):
    """shape filter per MODE: channel counts the bf16 patch kernel takes (multiples of 8) mixed with others; 3x3/s1 only for winograd; 1x1/s1/p0 for k1s"""
    rng = np.random.default_rng(seed + 7)
    out = []
    for sh in cases(40 * n, seed, big):
        b, c, h, w, oc, kh, kw, s, p = sh
        if mode == "winograd":
            sh = (b, c, h, w, oc, 3, 3, 1, min(p, 2))
            if h + 2 * sh[8] < 3 or w + 2 * sh[8] < 3: continue
        elif mode == "k1s":
            sh = (b, c, h, w, oc, 1, 1, 1, 0)
        elif mode == "bf16" and rng.random() < 0.6:
            sh = (b, int(rng.choice([16, 24, 32, 40, 64, 96])), h, w, oc, kh, kw, 1 if rng.random() < 0.7 else s, p)
            if (h + 2 * p - kh) < 0 or (w + 2 * p - kw) < 0: continue
        out.append(sh)
        if len(out) == n: break
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
    mode = os.environ.get("MODE", "exact")
    from boda_amd.digest import SsdsDiff
    rng = np.random.default_rng(seed + 13)
    worst = 0.0
    for sh in (cases(n, seed, big) if mode == "exact" else cases_mode(n, seed, big, mode)):
        op = _conv_op(*sh)
        spec = ""
        if mode == "winograd": rtc.set_tune("conv_algo", "winograd_all")
        if mode == "k1s":
            wi, wj = [(1, 4), (2, 2), (4, 1), (1, 8), (8, 1), (1, 1), (3, 1), (2, 4)][int(rng.integers(0, 8))]
            ocb = int(rng.integers(1, 5)); cb = int(rng.integers(1, 3))
            if (sh[1] + 1) // 2 * cb + 32 * ocb * cb + 30 > 256: cb = 1
            if (sh[1] + 1) // 2 * cb + 32 * ocb * cb + 30 > 256: ocb = 1
            spec = f"{wi}x{wj}x{ocb}x{cb}"
            if rng.random() < 0.5:   # the 16-bytes-per-lane kernel (k1_quad_f32.hip): qWJxOCBxRING, RING a divisor of the K step count; planes of >= 4 pels
                ks = (sh[1] + 1) // 2; divs = [d for d in range(1, 17) if ks % d == 0]
                spec = f"q{int(rng.choice([1, 2, 4, 8]))}x{int(rng.integers(1, 4))}x{int(rng.choice(divs))}" + (f"x{int(rng.integers(1, 3))}" if rng.random() < 0.5 else "")
            rtc.set_tune("k1_stream", spec)
        try:
            outs, prc = _run(be, op, 5, tune=OpTune(hip_tile=tile, hip_dtype="bf16" if mode == "bf16" else ""), include_ins=True)
        except Exception as e:
            if mode == "k1s" and "unsupported configuration" in str(e): modes["(spec refused)"] = modes.get("(spec refused)", 0) + 1; continue
            raise
        g = op.conv_geom()
        cfg = prc.launch["kernel"].replace("bodahip_", "") + " " + prc.launch["cfg"]; modes[cfg] = modes.get(cfg, 0) + 1
        if mode == "bf16":
            want = bo.conv_fwd(bo.to_bf16(outs["in"]), bo.to_bf16(outs["filts"]), outs["biases"], (g["SY"], g["SX"]), (g["PY"], g["PX"]), True)
        else:
            want = bo.conv_fwd(outs["in"], outs["filts"], outs["biases"], (g["SY"], g["SX"]), (g["PY"], g["PX"]), True)
        if mode in ("exact", "k1s"):
            ok = np.array_equal(want, outs["out"])
        else:
            sd = SsdsDiff.of(want, outs["out"]); worst = max(worst, sd.mrd); K = sh[1] * sh[5] * sh[6]   # fp32 accumulation-order noise on near-zero outputs grows like K (eps x partial-sum magnitude, K times): the stated bf16 bound is 1e-3 up to K = 2400
            ok = (not sd.has_nan()) and sd.mrd < (1e-3 * max(1.0, K / 2400.0) if mode == "bf16" else 2e-3)
        if mode == "k1s" and prc.launch["kernel"] != ("bodahip_k1_quad_f32" if spec.startswith("q") else "bodahip_k1_stream_f32"): ok = False
        if not ok:
            bad += 1; print("MISMATCH", sh, cfg, spec, int((want != outs["out"]).sum()), "of", want.size, flush=True)
    print(f"MODE={mode}: {n} cases, {bad} mismatches" + (f", worst mrd {worst:.2e}" if worst else "") + "; kernels / tile configs used:", dict(sorted(modes.items(), key=lambda kv: -kv[1])))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
