#!/bin/bash
# K hand-off with ONE persistent workgroup per CU (BODAHIP_KHO_OCC=1): no job ever waits for its predecessor
O=gpurun_out/kho1; mkdir -p $O
gen() { out=""; for t in $1; do for s in $2; do out="$out,${t}x$s"; done; done; echo "${out#,}"; }
P3="128x128x36x2x2x1x1x32x1x0 128x256x36x2x4x1x1x32x1x0 64x256x36x1x4x1x1x32x1x0 128x128x36x2x2x1x1x32x2x0 128x256x36x2x4x1x1x32x2x0"
K1="128x128x16x2x2x1x1x32x1x0 128x256x16x2x4x1x1x32x1x0 128x128x16x2x2x1x1x32x2x0 128x256x16x2x4x1x1x32x2x0 128x128x32x2x2x1x1x32x2x0"
export BODAHIP_KHO_OCC=1
for b in 128 256; do
  timeout 600 python tools/tile_sweep.py --workload nin --batch $b --ops 9 --tiles "$(gen "$P3" "4 8 12")" > $O/nin${b}_conv4.txt 2>&1
  timeout 600 python tools/tile_sweep.py --workload nin --batch $b --ops 10 --tiles "$(gen "$K1" "4 8 16")" > $O/nin${b}_cccp7.txt 2>&1
done
timeout 600 python tools/tile_sweep.py --workload alexnet --batch 256 --ops 4 --tiles "$(gen "$P3" "4 8")" > $O/alex256_conv5.txt 2>&1
grep -h "^op" $O/*.txt | cut -c1-170
