#!/bin/bash
# GPU tool: A/B of one environment switch on a bench.py workload, alternating, three repetitions.   tools/env_ab.sh "<ENV=VAL>" <bench.py args...>
cd "$(dirname "$0")/.."
E=$1; shift
run() { env $2 python bench.py "${@:3}" --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('%-8s' % '$1', 'value %.3f %s  ms/step %.4f  frac %.4f' % (d['value'], d['unit'], d['ms_per_step'], d['roofline']['frac']))
"; }
for i in 1 2 3; do run default "X=1" "$@"; run switched "$E" "$@"; done
