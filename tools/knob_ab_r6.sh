#!/bin/bash
# GPU tool: last-mile knobs of the staging-wave kernels, in place: LDS stages, tile -> XCD grouping, K step of the 1x1 forms
cd "$(dirname "$0")/.."; O=gpurun_out/knobs; mkdir -p $O
(
for e in BODAHIP_CBIG_NSTG=3 "BODAHIP_EXTRA_DEFS=-DGROUP_I=4" "BODAHIP_EXTRA_DEFS=-DGROUP_I=16"; do
  for w in "alexnet 256" "nin 256"; do echo "== $w $e (switched) vs default"; timeout 300 bash tools/env_ab_ops.sh "$e" $w 2; done
done
for T in 128x256x32x2x4x2x1x32x2x2 128x256x8x2x4x2x1x32x2x2; do echo "== nin 256 cccp5/6 -> $T"; timeout 200 bash tools/wisdom_ab.sh nin 256 "7=$T;8=$T" | tail -4; done
for T in 64x192x32x2x2x2x1x32x2x2 64x192x8x2x2x2x1x32x2x2; do echo "== nin 256 cccp7/8 -> $T"; timeout 200 bash tools/wisdom_ab.sh nin 256 "10=$T;11=$T" | tail -4; done
for i in 1 2; do for e in X=1 BODAHIP_EXTRA_DEFS=-DGROUP_I=4 BODAHIP_EXTRA_DEFS=-DGROUP_I=16; do env $e timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-conv-ops 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('%-32s %.2f TF/s %.3f ms |' % ('$e', d['value'], d['ms_per_step']), ' '.join('%.1f' % o['tflops'] for o in d['per_op'][9:]))
"; done; done
) 2>&1 | tee $O/log.txt
