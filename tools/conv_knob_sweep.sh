#!/bin/bash
# GPU tool: the big fp32 convolutions of the lists (AlexNet conv2-5 / NiN conv2-4 at 256 images) under the kernel's cheap knobs: workgroups per CU (MINW), MFMA-issue priority (SETPRIO),
# tile-walk group size (GROUP_I), store cache policy (ST_AUX).  Isolated launches (tools/tile_sweep.py): compare rows of one op with each other.
O=gpurun_out/knobs; mkdir -p $O
T="32x256x16x1x4x2,32x256x16x1x4x3,32x256x16x1x4x4,64x256x16x1x4x2,64x256x16x1x4x3,64x256x16x1x4x4,64x256x16x1x4x1"
for defs in "" "-DSETPRIO=1" "-DGROUP_I=4" "-DGROUP_I=16" "-DST_AUX=2"; do
  echo "== EXTRA_DEFS='$defs'"
  if [ -n "$defs" ]; then export BODAHIP_EXTRA_DEFS="$defs"; else unset BODAHIP_EXTRA_DEFS; fi
  python tools/tile_sweep.py --workload alexnet --batch 256 --ops 1,2,4 --iters 8 --tiles "$T" 2>&1 | grep "^op" | cut -c1-150
done > $O/alexnet.txt 2>&1
cat $O/alexnet.txt
