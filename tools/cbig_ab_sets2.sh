#!/bin/bash
# second round of in-sequence A/Bs: the two-workgroups-per-CU forms of the staging-wave kernel
S=x1x32x2x2
ab() { echo "### $1 $2 :: $3"; bash tools/wisdom_ab.sh $1 $2 "$3" 2>&1 | grep -v "^  File\|^Trace\|^Index" | tail -4; }
ab alexnet 256 "1=64x512x16x1x8x2$S;2=128x256x16x2x4x2$S;3=128x256x16x2x4x2$S;4=64x256x8x1x8x2$S"
ab alexnet 256 "1=128x128x16x2x4x2$S;2=64x512x16x1x8x2$S;3=64x512x16x1x8x2$S;4=256x128x16x4x2x2$S"
ab alexnet 256 "1=64x256x16x1x8x2$S;2=64x256x32x1x8x2$S;3=64x256x32x1x8x2$S;4=64x256x32x1x8x2$S;0=96x256x16x1x8x2$S"
ab nin 256 "3=128x128x16x2x4x2$S;6=128x256x16x2x4x2$S;4=128x128x16x2x4x2$S;5=128x128x16x2x4x2$S;9=64x256x8x1x8x2$S"
ab nin 256 "3=64x512x16x1x8x2$S;6=64x512x16x1x8x2$S;4=64x256x16x1x8x2$S;5=64x256x16x1x8x2$S;9=256x128x16x4x2x2$S;1=96x256x16x1x8x2$S;2=96x256x16x1x8x2$S"
ab nin 128 "3=64x512x16x1x8x2$S;6=128x256x16x2x4x2$S;4=128x128x16x2x4x2$S;5=128x128x16x2x4x2$S;7=64x256x16x1x8x2$S;8=64x256x16x1x8x2$S;9=128x128x16x2x4x2$S;10=128x128x16x2x4x2$S"
ab nin 128 "3=128x128x16x2x4x2$S;6=64x256x8x1x8x2$S;9=64x128x16x2x4x2$S;10=64x128x16x2x4x2$S;11=64x128x16x2x4x2$S;0=96x256x16x1x8x2$S"
