#!/bin/bash
# one-off PMC probe: rocprofv3 --pmc <counters...> on one bench workload; prints per-dispatch averages grouped by grid size
W=$1; shift
R=$PWD; O=$R/gpurun_out/pmc_probe; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc "$@" -d $O -o p -- python $R/bench.py --workload $W --steps 1 --warmup 0 --settle-ms 0 --no-cpu-baseline > $O/log 2>&1
cd $R
python - <<PY
import sqlite3, glob
db = glob.glob("$O/*.db")[0]
c = sqlite3.connect(db)
rows = c.execute("select grid_size, counter_name, sum(value), count(*), sum(end-start) from counters_collection where kernel_name like 'bodahip_%' group by grid_size, counter_name order by grid_size").fetchall()
cur = None
for g, cn, v, n, dur in rows:
    if g != cur: print(f"grid {g}  (dispatch-rows {n}, dur_sum {dur/1e3:.1f} us)"); cur = g
    print(f"    {cn:28s} {v:16.0f}")
PY
