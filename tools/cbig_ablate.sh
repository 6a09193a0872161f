#!/bin/bash
# Where does conv_big_f32.hip lose its time?  Ablations (-DABLATE: 1 no output stores | 2 no gather loads | 8 no filter loads) and structure knobs on three layers.
O=gpurun_out/cbig; mkdir -p $O
run() { # label, extra defs, env
  echo "== $1" ; BODAHIP_EXTRA_DEFS="$2" python tools/cbig_probe.py --time "$3" --ops "$4" --tiles "$5" --iters 6 2>&1 | grep -v "^$" ; }
T1=128x512x8x2x4x1x1x32x2x2; T2=256x256x16x2x4x1x1x32x2x2; T3=128x128x16x2x4x2x1x32x2x2
for ab in 0 1 2 8 10 11; do
  run "ABLATE=$ab alexnet conv3" "-DABLATE=$ab" alexnet:256 2 $T1,$T2,$T3
  run "ABLATE=$ab nin cccp5" "-DABLATE=$ab" nin:256 7 $T1,$T2,$T3
done
for pf in 1 4; do run "PF=$pf" "" alexnet:256 2 128x512x8x2x4x1x1x32x${pf}x2,256x256x16x2x4x1x1x32x${pf}x2; done
BODAHIP_CBIG_NSTG=3 run "NSTG=3" "" alexnet:256 2 $T1,$T2,256x256x32x2x4x1x1x32x2x2
