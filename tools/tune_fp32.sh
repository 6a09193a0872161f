#!/bin/bash
# Per-layer tile search for the exact fp32 lists (tools/tune_tiles.py = the reference's op-tuner idea): NiN at 128 / 256 images, AlexNet at 256.
#   gpurun --timeout 2400 -- 'bash tools/tune_fp32.sh'      -> gpurun_out/tune/<workload>_b<batch>.{txt,wis}
O=gpurun_out/tune; mkdir -p $O
C="64x64x16x2x2x2x1x32x2,64x64x32x2x2x2x1x32x2,64x64x16x2x2x2,64x64x32x2x2x2,128x128x16x2x2x2,128x128x32x2x2x2,128x128x16x2x2x2x1x32x2,128x128x32x2x2x2x1x32x2"
C="$C,64x256x16x1x4x2,64x256x32x1x4x2,64x256x16x1x4x2x1x32x2,128x256x16x2x4x1,128x256x32x2x4x1,128x256x16x2x4x1x1x32x2,32x256x16x1x4x2,32x256x32x1x4x2,32x128x16x1x4x2"
C="$C,64x128x16x1x4x2,64x128x32x1x4x2,64x128x16x2x2x2,128x64x16x2x2x2,128x64x32x2x2x2,256x64x16x4x1x2,256x128x16x4x2x1,96x128x16x1x4x2,96x256x16x1x4x2,32x64x32x2x4x1x1x16x2"
for wb in "nin 128" "nin 256" "alexnet 256"; do
  set -- $wb
  timeout 1500 python tools/tune_tiles.py --workload $1 --batch $2 --tiles "$C" --iters 14 --min-gain 0.025 --out $O/$1_b$2.wis > $O/$1_b$2.txt 2>&1
  tail -1 $O/$1_b$2.txt
done
