#!/bin/bash
# GPU tool: per-launch durations (rocprofv3 kernel trace, grouped by grid) of one decomposition of the sgemm list.   tools/sgemm_parts_trace.sh <label> "<BODAHIP_SGEMM_PARTS value>"
cd "$(dirname "$0")/.."; R=$PWD; O=$R/gpurun_out/sgemm_trace_$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BODAHIP_SGEMM_PARTS="$2" rocprofv3 --kernel-trace --stats -d $O/p -o p -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-conv-ops > $O/bench.log 2>&1
python $R/tools/rocprof_summary.py $(ls $O/p/*/*.db $O/p/*.db 2>/dev/null | head -1) --by-grid 2>&1 | grep -E "sgemm_big.*grid" > $O/by_grid.txt
cat $O/by_grid.txt
