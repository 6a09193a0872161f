#!/bin/bash
# GPU tool (round 6): hand-made guillotine decompositions of the ragged sizes of sgemm-ops-full (BODAHIP_SGEMM_PARTS) against the planner's choice, in the layer sequence.
cd "$(dirname "$0")/.."; O=gpurun_out/sgemm_parts; mkdir -p $O
N=${1:-2}
run() { env "$2" python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-conv-ops 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('%-10s' % '$1', '%.2f TF/s %.3f ms |' % (d['value'], d['ms_per_step']), ' '.join('%.1f' % o['tflops'] for o in d['per_op'][10:]))
"; }
W=256x128x8x3x4x1; Q=256x256x16x2x4x1x1x32x2; S=64x64x16x2x2x4x1x32x2x3; T=128x128x8x3x4x2
P1="6144:0,5376,0,6144,$W/5376,768,0,6144,$S;7168:0,6912,0,7168,$W/6912,256,0,7168,$S"
P2="6144:0,6144,0,5376,$W/0,6144,5376,768,$S;7168:0,7168,0,6912,$W/0,7168,6912,256,$S"
( echo "sizes: 4096 5120 6144 7168 8192 10240 12288"
for i in $(seq $N); do
  run base X=1
  for v in P1 P2; do run $v "BODAHIP_SGEMM_PARTS=${!v}"; done
done ) 2>&1 | tee $O/log.txt
