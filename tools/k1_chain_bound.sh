#!/bin/bash
# GPU tool (round-5 verdict item 5): what could a fused cccp3 -> cccp4 launch (NiN, 256 -> 256 channels at 27 x 27, 128 images) save at most?  The chain removes the write of
# the intermediate tensor from cccp3 and its read from cccp4 -- so time the two layers on the staging-wave kernel with exactly those accesses ablated
# (-DABLATE=1: no output stores; 2: no pel loads): the gap to the unablated launch is the ceiling of the saving, before the chain's own costs (64-pel tiles, both filters
# re-streamed per tile) are counted.
O=gpurun_out/k1bound; mkdir -p $O
run() { echo "== $1"; BODAHIP_EXTRA_DEFS="$2" python tools/cbig_probe.py --time "$3" --ops "$5" --tiles "$4" --iters 8 2>&1 | grep -v "^$"; }
T=128x128x16x2x2x2x1x32x2x2,128x128x16x2x4x2x1x32x2x2,64x256x16x1x8x2x1x32x2x2
( for ab in 0 1 2 0 1 2; do run "ABLATE=$ab nin@128 cccp3 (op 4)" "-DABLATE=$ab" nin:128 $T 4; done ) 2>&1 | tee $O/log.txt
