#!/bin/bash
# GPU tool: LLVM AMDGPU scheduler strategies for the native kernels (extra compiler options through BODAHIP_EXTRA_DEFS), AlexNet list in place
cd "$(dirname "$0")/.."; O=gpurun_out/sched; mkdir -p $O
run() { BODAHIP_EXTRA_DEFS="$2" python bench.py --workload ${W:-alexnet} --batch ${B:-256} --steps 10 --warmup 3 --no-cpu-baseline 2>$O/err.txt | python -c "
import sys, json
l=[l for l in sys.stdin if l.startswith('{')]
if not l: print('%-44s FAILED' % '$1'); sys.exit(0)
d = json.loads(l[0])
print('%-44s' % '$1', 'ms/step %.4f  frac %.4f |' % (d['ms_per_step'], d['roofline']['frac']), ' '.join('%.0f' % (o['ms'] * 1e3) for o in d['per_op']))
"; }
( for i in 1 2; do
  run "noop (default options)" "-DNOOP=1"
  for s in max-ilp max-memory-clause iterative-ilp iterative-minreg; do run "sched-strategy=$s" "-DNOOP=1 -mllvm -amdgpu-sched-strategy=$s"; done
  run "misched=gcn-max-occupancy-experimental" "-DNOOP=1 -mllvm -misched=gcn-max-occupancy-experimental"
  run "-O2" "-DNOOP=1 -O2"
done ) 2>&1 | tee $O/log.txt
