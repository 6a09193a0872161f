#!/bin/bash
# GPU tool: config-5 lists as edge-free graphs -- plain, implicit-GEMM members as one multi-problem launch, and sets of specialised members of several sizes
cd "$(dirname "$0")/.."
run() { python bench.py "$@" --dtype bf16 --layout nhwc --graph --independent --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('  ms/step %.4f  wall frac %s  launches %s' % (d['ms_per_step'], d['roofline']['timed_region']['frac'], d['roofline']['timed_region']['launches_per_step']))"; }
for w in googlenet resnet50; do for v in "" "--multi" "--sets 4" "--sets 8" "--sets 16" "--multi --sets 8"; do echo "$w $v:"; run --workload $w $v; done; done
