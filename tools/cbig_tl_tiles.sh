#!/bin/bash
# K-loop time of conv_big_f32.hip from its clock stamps under several tile strings:  cbig_tl_tiles.sh "<spec> <op>" tile...
SP=$1; shift
for t in "$@"; do
  echo "== $t"; python tools/cbig_timeline.py --spec ${SP% *} --op ${SP#* } --tile $t 2>&1 | grep -A7 "^launch 1" | grep "K loop"
done
