#!/bin/bash
# GPU tool: A/B of one environment switch on a conv-ops workload with the per-op times, alternating.   tools/env_ab_ops.sh "<ENV=VAL>" <workload> <batch> [reps]
cd "$(dirname "$0")/.."
E=$1; W=$2; B=$3; N=${4:-2}
run() { env $2 python bench.py --workload $W --batch $B --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('%-9s' % '$1', 'ms/step %.4f  frac %.4f |' % (d['ms_per_step'], d['roofline']['frac']), ' '.join('%.0f' % (o['ms'] * 1e3) for o in d['per_op']))
"; }
for i in $(seq $N); do run default "X=1"; run switched "$E"; done
