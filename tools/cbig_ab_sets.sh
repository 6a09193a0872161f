#!/bin/bash
# in-sequence A/Bs (tools/wisdom_ab.sh) of staging-wave tiles per layer: candidate sets per workload
S=x1x32x2x2
ab() { echo "### $1 $2 :: $3"; bash tools/wisdom_ab.sh $1 $2 "$3" 2>&1 | tail -6; }
ab alexnet 256 "1=64x256x8x1x8x2$S;2=64x256x8x1x8x2$S;3=64x256x8x1x8x2$S;4=128x128x16x2x4x2$S"
ab alexnet 256 "1=64x512x16x1x8x1$S;2=128x256x16x2x4x1$S;3=64x512x16x1x8x1$S;4=96x256x16x1x8x1$S"
ab alexnet 256 "1=128x256x32x2x4x1$S;2=64x512x16x1x8x1$S;3=128x256x16x2x4x1$S;4=128x384x16x2x4x1$S"
ab nin 256 "3=64x256x8x1x8x2$S;6=64x256x8x1x8x2$S;7=128x512x16x2x4x1$S;8=128x512x16x2x4x1$S;9=128x384x16x2x4x1$S;10=192x256x16x2x4x1$S;11=192x256x16x2x4x1$S"
ab nin 256 "3=64x512x16x1x8x1$S;6=128x256x16x2x4x1$S;7=128x128x16x2x4x2$S;8=128x128x16x2x4x2$S;9=96x256x16x1x8x1$S;10=256x192x16x4x2x1$S;11=256x192x16x4x2x1$S"
ab nin 128 "3=64x512x16x1x8x1$S;6=128x128x16x2x4x2$S;7=128x128x16x2x4x2$S;8=128x128x16x2x4x2$S;9=96x256x16x1x8x1$S;10=96x256x16x1x8x1$S;11=96x256x16x1x8x1$S"
ab nin 128 "3=64x256x8x1x8x2$S;6=128x256x16x2x4x1$S;7=64x256x8x1x8x2$S;8=64x256x8x1x8x2$S;9=64x256x8x1x8x2$S;4=256x192x16x4x2x1$S;5=256x192x16x4x2x1$S"
