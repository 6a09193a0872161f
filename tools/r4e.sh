#!/bin/bash
mkdir -p gpurun_out/r4e; O=gpurun_out/r4e
python -m pytest tests/test_gpu_nhwc.py -m gpu -q -p no:cacheprovider -x -k "multi_problem or edge_free" > $O/pytest.log 2>&1; echo rc=$? >> $O/pytest.log; tail -30 $O/pytest.log
run() { # name, env..., args
  local nm=$1; shift
  python bench.py --dtype bf16 --layout nhwc --graph --steps 30 --warmup 5 --no-cpu-baseline "$@" 2>$O/err_$nm.log | tail -1 | python -c "
import sys,json
try:
  d=json.loads(sys.stdin.read()); print('$nm', d['value'], d['ms_per_step'], d['roofline']['timed_region']['frac'], d['config'].get('multi_problem_members'), d['config'].get('launch'))
except Exception as e: print('$nm', 'FAILED', e)" >> $O/lists.txt 2>&1
  tail -2 $O/err_$nm.log >> $O/lists.txt
}
for wl in googlenet resnet50; do
  run ${wl}_ind --workload $wl --independent
  run ${wl}_multi_chain --workload $wl --multi
  run ${wl}_multi_ind --workload $wl --independent --multi
  BENCH_MULTI_MAX_TILES=400 run ${wl}_multi_ind_max400 --workload $wl --independent --multi
  BENCH_MULTI_MAX_TILES=1000 run ${wl}_multi_ind_max1000 --workload $wl --independent --multi
  BENCH_MULTI_TILE=64x128x64x2x2x2 run ${wl}_multi_ind_t64x128 --workload $wl --independent --multi
  BENCH_MULTI_TILE=128x128x64x2x2x2 run ${wl}_multi_ind_t128x128 --workload $wl --independent --multi
  BENCH_MULTI_TILE=32x128x64x1x4x2 run ${wl}_multi_ind_t32x128 --workload $wl --independent --multi
  BENCH_MULTI_TILE=64x128x32x2x2x2x1x32x4 run ${wl}_multi_ind_t64x128x32n4 --workload $wl --independent --multi
done
cat $O/lists.txt
