#!/usr/bin/env python3
"""Staging-wave fp32 convolution kernel (kernels/conv_big_f32.hip): bit-exactness against the oracle on edge shapes under every tile form, then layer timings
against the planner's own choice.   usage: cbig_probe.py [--check] [--time alexnet:256,nin:256,nin:128] [--tiles t1,t2] [--ops 1,2] [--iters 8]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from boda_amd.cnn_op import OpTune, add_codegen_annotations
from boda_amd.op import parse_op
from boda_amd.ops_prof import OpsBackend, profile_rcg_call
from boda_amd.rtc import make_rtc
import bench

TILES = ["256x256x16x2x4x1x1x32x2x2", "256x256x32x2x4x1x1x32x2x2", "256x128x16x4x2x1x1x32x2x2", "128x256x16x2x4x1x1x32x2x2", "128x256x32x2x4x1x1x32x2x2", "128x128x16x2x4x2x1x32x2x2", "96x256x16x1x8x1x1x32x2x2",
         "96x512x8x1x8x1x1x32x2x2", "256x192x16x4x2x1x1x32x2x2", "192x256x16x2x4x1x1x32x2x2", "128x512x8x2x4x1x1x32x2x2", "128x512x16x2x4x1x1x32x2x2", "64x256x8x1x8x2x1x32x2x2", "128x384x8x2x4x1x1x32x2x2", "128x384x16x2x4x1x1x32x2x2", "64x512x16x1x8x1x1x32x2x2"]


def conv_op(B, C, H, W, OC, KH, KW, S, P):
    OH = (H + 2 * P - KH) // S + 1; OW = (W + 2 * P - KW) // S + 1
    return parse_op(f"(str_vals=(type=Convolution),nda_vals=(biases=(dims=(out_chan={OC})),filts=(dims=(out_chan={OC},in_chan={C},y={KH},x={KW})),"
                    f"in=(dims=(img={B},chan={C},y={H},x={W})),in_pad=(tn=none,dims=(y={P},x={P})),kern_sz=(tn=none,dims=(y={KH},x={KW})),"
                    f"out=(dims=(img={B},chan={OC},y={OH},x={OW})),out_chans=(tn=uint32_t,v={OC}),stride=(tn=none,dims=(y={S},x={S}))))")


def check(be, tiles):
    from oracle import boda_oracle as bo
    shapes = [(3, 24, 15, 15, 100, 3, 3, 1, 1), (2, 3, 35, 35, 96, 11, 11, 4, 0), (5, 96, 7, 7, 130, 1, 1, 1, 0), (4, 17, 9, 9, 70, 5, 5, 1, 2), (7, 16, 6, 6, 100, 6, 6, 1, 0),
              (3, 33, 13, 13, 33, 1, 1, 2, 0), (2, 10, 12, 12, 300, 3, 3, 2, 1), (9, 20, 1, 1, 50, 1, 1, 1, 0), (1, 4, 40, 40, 8, 3, 3, 1, 1)]
    bad = 0
    for sh in shapes:
        op = conv_op(*sh)
        for t in tiles:
            try:
                tune = OpTune(hip_tile=t)
                anno = add_codegen_annotations(op, tune)
                outs, prc = profile_rcg_call(be, anno, 5, 0.0, 1, include_ins=True, tile=t)
                want = bo.conv_fwd(outs["in"], outs["filts"], outs["biases"], (sh[7], sh[7]), (sh[8], sh[8]), True)
                ok = np.array_equal(want, outs["out"])
                nbad = int(np.sum(want != outs["out"]))
                print(f"shape {sh} tile {t:>28s} [{prc.launch['cfg']}] {prc.launch['kernel']} {'OK' if ok else 'MISMATCH ' + str(nbad) + ' of ' + str(want.size)}", flush=True)
                bad += (not ok)
            except Exception as e:
                print(f"shape {sh} tile {t} ERR {type(e).__name__}: {str(e)[:160]}", flush=True); bad += 1
    print("CHECK", "PASS" if not bad else f"FAIL ({bad})", flush=True)
    return bad


def time_layers(be, spec, tiles, sel, iters):
    wl, batch = spec.split(":"); batch = int(batch)
    ops = {"alexnet": lambda: bench.alexnet_b256_ops(batch), "nin": lambda: bench.nin_ops(batch)}[wl]()
    for i, op in enumerate(ops):
        if sel and i not in sel: continue
        g = op.conv_geom()
        if g["OH"] == 1 and g["OW"] == 1 and g["KH"] > 1: continue
        res = []
        for t in [""] + tiles:
            try:
                anno = add_codegen_annotations(op, OpTune(hip_tile=t))
                _, prc = profile_rcg_call(be, anno, 5, run_iter=iters, want_outs=False, tile=t)
                best = min(prc.all_secs[1:]) if len(prc.all_secs) > 1 else prc.all_secs[0]
                res.append((best, t or "auto", prc.launch["cfg"], prc.launch["grid"]))
            except Exception as e:
                res.append((9e9, t, f"ERR {type(e).__name__}: {str(e)[:60]}", 0))
        base = res[0][0]
        print(f"{wl}@{batch} op {i:2d} C{g['C']} {g['H']}x{g['W']} OC{g['OC']} k{g['KH']}s{g['SY']}p{g['PY']}  auto [{res[0][2]}] grid {res[0][3]} {base*1e6:8.1f} us {op.flops()/base/1e12:6.1f} TF/s", flush=True)
        for b, t, cfg, grid in sorted(res[1:]):
            if b > 1e9: print(f"      {t:>30s} {cfg}", flush=True); continue
            print(f"      {t:>30s} grid {grid:5d} {b*1e6:8.1f} us {op.flops()/b/1e12:6.1f} TF/s  x{base/b:5.3f}", flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--time", default="")
    ap.add_argument("--tiles", default="")
    ap.add_argument("--ops", default="")
    ap.add_argument("--iters", type=int, default=8)
    a = ap.parse_args()
    tiles = [t for t in a.tiles.split(",") if t] or TILES
    rtc = make_rtc(); rtc.init(); be = OpsBackend(rtc)
    rc = 0
    if a.check: rc = check(be, tiles)
    sel = [int(x) for x in a.ops.split(",")] if a.ops else []
    for spec in [s for s in a.time.split(",") if s]:
        time_layers(be, spec, tiles, sel, a.iters)
    sys.exit(1 if rc else 0)
