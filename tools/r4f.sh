#!/bin/bash
# the multi-problem launch by itself: its event-timed duration against the sum / max of its members' own launches
mkdir -p gpurun_out/r4f; O=gpurun_out/r4f
probe() { # name, args...
  local nm=$1; shift
  python bench.py --dtype bf16 --layout nhwc --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>$O/err_$nm.log | tail -1 > $O/$nm.json
  python - <<P >> $O/probe.txt
import json
d=json.load(open("$O/$nm.json"))
po=d["per_op"]; big=max(po,key=lambda p:p["flops"])
print("$nm", "ops", len(po), "sum_ms", round(sum(p["ms"] for p in po),4), "members", d["config"].get("multi_problem_members"), "largest-op(ms,tflops,gbs)", big["ms"], big["tflops"], big["gbs"], "wall", d["ms_per_step"])
P
}
for wl in googlenet resnet50; do
  probe ${wl}_single --workload $wl
  probe ${wl}_multi --workload $wl --multi
  BENCH_MULTI_MAX_TILES=400 probe ${wl}_multi_max400 --workload $wl --multi
  BENCH_MULTI_TILE=64x128x32x2x2x2x1x32x4 probe ${wl}_multi_bk32n4 --workload $wl --multi
  BENCH_MULTI_TILE=64x128x64x2x2x3x1x32x2 probe ${wl}_multi_minw3n2 --workload $wl --multi
  BENCH_MULTI_TILE=64x64x64x2x2x2x1x32x3 probe ${wl}_multi_64x64 --workload $wl --multi
  BENCH_MULTI_TILE=64x256x32x2x2x2x1x32x3 probe ${wl}_multi_64x256 --workload $wl --multi
done
cat $O/probe.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/prof -o p -- python $OLDPWD/bench.py --workload googlenet --dtype bf16 --layout nhwc --multi --steps 5 --warmup 2 --no-cpu-baseline > $OLDPWD/$O/prof.log 2>&1
cd $OLDPWD; python tools/rocprof_summary.py $(find $O/prof -name "*.db" | head -1) --by-grid 2>&1 | head -40 > $O/prof_summary.txt; cat $O/prof_summary.txt; find $O -name "*.db" -size +20M -delete
