#!/bin/bash
# GPU tool (round 6): the staging-wave sgemm kernel's small forms (tile field 10 == 3) on the mid sizes of sgemm-ops-full and as the rest launch of the two-level tiling,
# in the layer sequence (bench.py default workload), alternating with the planner's own choice.   tools/sgemm_stg_ab.sh [reps]
cd "$(dirname "$0")/.."; O=gpurun_out/sgemm_stg; mkdir -p $O
N=${1:-2}
run() { env $2 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-conv-ops 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('%-10s' % '$1', '%.2f TF/s %.3f ms |' % (d['value'], d['ms_per_step']), ' '.join('%.1f' % o['tflops'] for o in d['per_op'][5:]))
"; }
mid() { echo "BODAHIP_SGEMM_TILE_FOR=768=$1;1024=$1;1536=$1;2048=$1;3072=$1"; }
A=64x64x16x2x2x4x1x32x2x3; B=64x128x16x2x2x4x1x32x2x3; C=128x128x8x2x2x3x1x32x2x3; D=128x128x8x3x4x2; E=128x64x16x2x2x4x1x32x2x3; F=64x128x16x1x4x4x1x32x2x3; G=128x128x16x2x2x3x1x32x2x3; H=64x256x16x1x8x2x1x32x2x3
( echo "sizes: 768 1024 1536 2048 3072 4096 5120 6144 7168 8192 10240 12288"
for i in $(seq $N); do
  run base X=1
  for v in A B C D E F G H; do run mid-$v "$(mid ${!v})"; done
  for v in A B F; do run tail-$v "BODAHIP_SGEMM_TAIL64=${!v} BODAHIP_SGEMM_SPLIT_TAIL=64"; done
done ) 2>&1 | tee $O/log.txt
