#!/bin/bash
mkdir -p gpurun_out/r4c; O=gpurun_out/r4c
for wl in googlenet resnet50; do
  for ind in "" "--independent"; do
    python bench.py --workload $wl --dtype bf16 --layout nhwc --graph $ind --steps 30 --warmup 5 --no-cpu-baseline 2>$O/err_${wl}_${ind}.log | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$wl', '$ind', d['value'], d['ms_per_step'], d['roofline'].get('frac'), d['config'].get('launch'))" >> $O/lists.txt 2>&1
  done
done
cat $O/lists.txt
python -m pytest tests -m gpu -q --durations=40 -p no:cacheprovider -x > $O/pytest.log 2>&1; echo rc=$? >> $O/pytest.log; tail -45 $O/pytest.log
