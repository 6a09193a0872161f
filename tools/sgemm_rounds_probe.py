#!/usr/bin/env python3
"""How long does a launch of n tiles of the staging-wave sgemm kernel take, n = 1 .. 6 rounds of the CUs and in between?  Round 6 found a 768-tile launch (three whole
rounds of 256 x 128 tiles) taking FOUR rounds' time.  One tile form, K fixed, M x N chosen for the tile count; several launches back to back, the fastest of the later
ones reported.   usage: python tools/sgemm_rounds_probe.py [tile] [K]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from boda_amd.cnn_op import OpTune, add_codegen_annotations
from boda_amd.op import parse_op
from boda_amd.ops_prof import OpsBackend, profile_rcg_call
from boda_amd.rtc import make_rtc
tile = sys.argv[1] if len(sys.argv) > 1 else "256x128x8x3x4x1"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 6144
bi, bj = [int(x) for x in tile.split("x")[:2]]
rtc = make_rtc(); rtc.init(); be = OpsBackend(rtc)
def sg(M, N): return parse_op(f"(str_vals=(type=sgemm),nda_vals=(a=(dims=(K={K},M={M})),b=(dims=(K={K},N={N})),c=(dims=(M={M},N={N}))))")
for ti, tj in [(16, 16), (16, 24), (16, 32), (16, 40), (16, 48), (24, 32), (32, 24), (12, 64), (48, 16), (16, 56), (16, 64), (32, 32), (16, 80), (40, 32), (16, 96), (24, 64), (26, 30), (28, 28)]:
    anno = add_codegen_annotations(sg(ti * bi, tj * bj), OpTune(hip_tile=tile))
    outs, prc = profile_rcg_call(be, anno, 5, 0.0, 8, tile=tile)
    best = min(prc.all_secs[2:])
    n = ti * tj
    print(f"{tile} K {K}: {ti:3d} x {tj:3d} = {n:5d} tiles = {n / 256:5.2f} rounds: {best * 1e3:8.3f} ms = {best * 1e3 / (n / 256):7.3f} ms per round-equivalent, {2.0 * ti * bi * tj * bj * K / best / 1e12:6.1f} TF/s", flush=True)
