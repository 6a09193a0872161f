#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite output) run into text: per-kernel stats (== --stats) and, when the run
collected PMC counters, per-kernel / per-grid counter sums.   usage: rocprof_summary.py results.db [--by-grid]"""
import sqlite3
import sys
from collections import defaultdict


def main():
    db = sys.argv[1]
    by_grid = "--by-grid" in sys.argv
    c = sqlite3.connect(db)
    print(f"# rocprofv3 summary of {db}")
    print("## kernel stats (ns)  [name, calls, total_ns, avg_ns, min_ns, max_ns, pct]")
    rows = list(c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by sum(duration) desc"))
    tot = sum(r[2] for r in rows) or 1
    for n, k, s, a, mn, mx in rows:
        print(f"{n:44s} {k:6d} {s:14.0f} {a:14.1f} {mn:12.0f} {mx:12.0f} {100.0*s/tot:7.2f}%")
    if by_grid:
        print("## per (kernel, grid, workgroup) [calls, avg_ns, min_ns, vgpr, accum_vgpr, sgpr, lds]")
        for r in c.execute("select name, grid_x, workgroup_x, count(*), avg(duration), min(duration), max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size) "
                           "from kernels group by name, grid_x, workgroup_x order by name, grid_x"):
            print(f"{r[0]:36s} grid {r[1]:9d} wg {r[2]:4d} calls {r[3]:4d} avg {r[4]:13.1f} min {r[5]:12.0f} vgpr {r[6]} agpr {r[7]} sgpr {r[8]} lds {r[9]}")
    if by_grid:
        # steady state (round 6): per (kernel, grid) the FIRST launch -- module load, cold caches: one pass in the run -- is dropped, the rest averaged; the number of passes
        # over the op list is the launch count of the rarest (kernel, grid); per-step total = sum over groups of avg x launches per pass
        rows = list(c.execute("select name, grid_x, start, duration from kernels order by start"))
        groups = {}
        for n, g, st, du in rows: groups.setdefault((n, g), []).append(du)
        big = {k: v for k, v in groups.items() if not k[0].startswith(("gen_data", "bodahip_gen", "calib"))}
        if big:
            # passes over the op list = the launch count that carries the most kernel time (set-up kernels -- data generation, one-time layout passes -- run once and are left out)
            by_count = {}
            for v in big.values(): by_count[len(v)] = by_count.get(len(v), 0) + sum(v)
            passes = max(by_count, key=by_count.get)
            big = {k: v for k, v in big.items() if len(v) >= passes}
            tot = 0.0
            print(f"## steady state: first launch of every (kernel, grid) dropped; {passes} passes over the op list  [name, grid, launches, avg_ns of the rest, per pass]")
            for (n, g), v in sorted(big.items()):
                rest = v[1:] if len(v) > 1 else v
                avg = sum(rest) / len(rest); per = len(v) / passes
                tot += avg * per
                print(f"{n:44s} grid {g:9d} n {len(v):5d} avg {avg:13.1f} x {per:5.2f}")
            print(f"## steady-state kernel time per step (one pass over the op list): {tot / 1e6:.4f} ms")
    try:
        pm = list(c.execute("select kernel_name, grid_size, counter_name, count(*), sum(value), avg(value) from counters_collection group by kernel_name, grid_size, counter_name order by kernel_name, grid_size"))
    except sqlite3.Error:
        pm = []
    if pm:
        print("## PMC counters per (kernel, grid): [counter, dispatches, sum, avg]")
        for kn, g, cn, n, s, a in pm:
            print(f"{kn:36s} grid {g:9d} {cn:28s} n {n:4d} sum {s:18.2f} avg {a:16.2f}")


if __name__ == "__main__":
    main()
