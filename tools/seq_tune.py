#!/usr/bin/env python3
"""IN-SEQUENCE per-layer tile search (round 5): one bench.py run of the whole list per (layer, candidate tile), the tile handed to that layer through the tile-wisdom path
(BODAHIP_TILE_WISDOM); a base run (the planner's plan) after every few candidates.  Isolated-launch sweeps (tools/tile_sweep.py, tools/tune_tiles.py) run at other clocks
and cache states and have put tiles 8-13 % ahead that lose in the layer sequence -- only these numbers decide a plan.
usage: seq_tune.py <alexnet|nin> <batch> [op,op,...]        -> one line per layer: base us | tile us ... (best marked)"""
import json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
w, b = sys.argv[1], int(sys.argv[2])
ops = bench.alexnet_b256_ops(b) if w == "alexnet" else bench.nin_ops(b)
sel = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else None
PATCH = ["32x256x16x1x4x2", "64x256x16x1x4x2", "64x256x32x1x4x2", "128x256x16x2x4x1", "64x64x32x2x2x2x1x32x2", "128x128x16x2x2x2", "64x128x16x1x4x2", "32x128x16x1x4x2", "128x128x32x2x2x2"]
K1 = ["64x64x16x2x2x2x1x32x2", "64x64x32x2x2x2x1x32x2", "128x128x16x2x2x2", "128x128x32x2x2x2", "128x64x16x2x2x2", "64x128x16x1x4x2", "128x256x16x2x4x1", "256x128x16x4x2x1"]
def run(wis):
    env = dict(os.environ)
    if wis: env["BODAHIP_TILE_WISDOM"] = wis
    else: env.pop("BODAHIP_TILE_WISDOM", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", w, "--batch", str(b), "--steps", "10", "--warmup", "3", "--no-cpu-baseline"], capture_output=True, text=True, env=env, cwd=ROOT)
    line = next((l for l in r.stdout.splitlines() if l.startswith("{")), None)
    return [o["ms"] * 1e3 for o in json.loads(line)["per_op"]] if line else None
for i, op in enumerate(ops):
    g = op.conv_geom()
    if (sel is not None and i not in sel) or (g["OH"] == 1 and g["OW"] == 1) or g["KH"] > 7: continue      # (fully-connected layers and conv1 have planners of their own)
    cands = K1 if g["KH"] == 1 else PATCH
    base, res = [], []
    for n, t in enumerate(cands):
        if n % 3 == 0:
            r = run(None); base.append(r[i] if r else float("nan"))
        with tempfile.NamedTemporaryFile("w", suffix=".wis", delete=False) as f:
            f.write(f"{op.to_str()}\t{t}\t0\t0\n")
        r = run(f.name); os.unlink(f.name)
        res.append((t, r[i] if r else float("nan")))
    r = run(None); base.append(r[i] if r else float("nan"))
    b_us = sorted(base)[len(base) // 2]; best = min(res, key=lambda x: x[1])
    print(f"op {i:2d} C{g['C']:4d} {g['H']:3d}x{g['W']:<3d} OC{g['OC']:4d} k{g['KH']}  base {b_us:7.1f} us (runs {' '.join('%.0f' % x for x in base)}) | " +
          "  ".join(f"{t} {us:.1f}{'*' if (t, us) == best and us < b_us * 0.985 else ''}" for t, us in res), flush=True)
