#!/bin/bash
# K-loop time of conv_big_f32.hip from its clock stamps under ablations of the staging waves' work (alexnet conv3, 128x512 tile)
for ab in ${ABS:-0 1 2 4 8 10 14}; do
  echo "== ABLATE=$ab"; BODAHIP_EXTRA_DEFS="-DABLATE=$ab" python tools/cbig_timeline.py ${1:-} 2>&1 | grep -A9 "^launch 1" | grep "K loop\|barriers\|staging"
done
