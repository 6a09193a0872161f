#!/bin/bash
# Rolling-rows kernel: waves per workgroup x row runs per image, GoogLeNet net @64 (graph replay) and the stem call's own time (call by call, backend events).
O=gpurun_out/rows_sweep; mkdir -p $O
G="--workload googlenet-net --dtype bf16 --layout nhwc --no-cpu-baseline"
for cfg in "8 0" "4 0" "8 8" "4 8" "8 2" "4 16"; do
  set -- $cfg; export BODAHIP_NHWC_ROWS_WJ=$1; if [ $2 = 0 ]; then unset BODAHIP_NHWC_ROWS_CHUNKS; else export BODAHIP_NHWC_ROWS_CHUNKS=$2; fi
  for mode in fused apart; do
    X=""; [ $mode = apart ] && X="--no-fuse-post"
    timeout 300 python bench.py $G --graph --steps 30 --warmup 5 $X > $O/g_${1}_${2}_$mode.json 2> $O/g_${1}_${2}_$mode.err
    timeout 300 python bench.py $G --per-op --timing kernel --steps 20 --warmup 5 $X > $O/p_${1}_${2}_$mode.json 2> $O/p_${1}_${2}_$mode.err
    echo "WJ=$1 chunks=$2 $mode: $(python -c "import json,sys; j=json.loads(open('$O/g_${1}_${2}_$mode.json').read().strip().splitlines()[-1]); print(j['ms_per_step'], j['images_per_s'])" 2>&1 | tail -1) | $(grep -E "^\s+conv1" $O/p_${1}_${2}_$mode.err | head -1)"
  done
done
unset BODAHIP_NHWC_ROWS_WJ BODAHIP_NHWC_ROWS_CHUNKS
BODAHIP_NHWC_ROWS=0 timeout 300 python bench.py $G --graph --steps 30 --warmup 5 --no-fuse-post > $O/g_patch.json 2> $O/g_patch.err
BODAHIP_NHWC_ROWS=0 timeout 300 python bench.py $G --per-op --timing kernel --steps 20 --warmup 5 --no-fuse-post > $O/p_patch.json 2> $O/p_patch.err
echo "patch: $(python -c "import json,sys; j=json.loads(open('$O/g_patch.json').read().strip().splitlines()[-1]); print(j['ms_per_step'], j['images_per_s'])" 2>&1 | tail -1)"; grep -E "^\s+(conv1|pool1)" $O/p_patch.err | head -3
