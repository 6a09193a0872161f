#!/usr/bin/env python3
"""Random-shape check of the sgemm planner's decompositions (round 6: guillotine cuts, row split, staging-wave small forms): M, N multiples of 64 (some of 4) in 1024 .. 12288,
K in 512 .. 1100 -- the planned launch list against ONE launch of the general kernel's 128 x 128 tile, bit for bit, and against the oracle where the problem is small.
usage: fuzz_sgemm_parts.py [n_cases] [seed]   (GPU box; exit code 1 on any mismatch)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from boda_amd import rtc as R
from boda_amd.cnn_op import OpTune, add_codegen_annotations
from boda_amd.op import parse_op
from boda_amd.ops_prof import OpsBackend, profile_rcg_call
from oracle import boda_oracle as bo
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
rtc = R.make_rtc(); rtc.init(); be = OpsBackend(rtc)
def sg(M, N, K): return parse_op(f"(str_vals=(type=sgemm),nda_vals=(a=(dims=(K={K},M={M})),b=(dims=(K={K},N={N})),c=(dims=(M={M},N={N}))))")
bad = 0; kinds = {}
for i in range(n):
    g = 64 if rng.random() < 0.8 else 4
    M = int(rng.integers(1024 // g, 12288 // g + 1)) * g; N = int(rng.integers(1024 // g, 12288 // g + 1)) * g; K = int(rng.integers(512, 1101))
    if M * N > 90e6: N = max(1024, int(90e6 / M) // g * g)
    op = sg(M, N, K)
    plan = R.explain_plan(op); kind = plan.split()[0] if plan.startswith(("parts=", "rows<")) else plan.split()[1]
    kind = "parts" if plan.startswith("parts=") else "rows" if plan.startswith("rows<") else plan.split()[0].replace("bodahip_", "") + " " + plan.split()[1]
    kinds[kind] = kinds.get(kind, 0) + 1
    got, prc = profile_rcg_call(be, add_codegen_annotations(op, OpTune()), 5, 0.0, 1, include_ins=(M * N * K < 3e9))
    ref, _ = profile_rcg_call(be, add_codegen_annotations(op, OpTune(hip_tile="128x128x16x2x2x2")), 5, 0.0, 1, tile="128x128x16x2x2x2")
    ok = np.array_equal(got["c"], ref["c"])
    if ok and "a" in got: ok = np.array_equal(bo.sgemm(got["a"], got["b"]), got["c"])
    if not ok: bad += 1; print(f"MISMATCH M {M} N {N} K {K}: {plan[:200]}", flush=True)
print(f"fuzz_sgemm_parts: {n} cases, {bad} mismatches; plans: {kinds}")
sys.exit(1 if bad else 0)
