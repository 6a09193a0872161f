#!/bin/bash
# A/B of the two-level sgemm tiling on one box: per-size TF/s for the unsplit planner, split with 128x128 / 64x64 tails, automatic
for mode in nosplit t128 t64 auto; do
  unset BODAHIP_NO_SGEMM_SPLIT BODAHIP_SGEMM_SPLIT_TAIL
  case $mode in nosplit) export BODAHIP_NO_SGEMM_SPLIT=1;; t128) export BODAHIP_SGEMM_SPLIT_TAIL=128;; t64) export BODAHIP_SGEMM_SPLIT_TAIL=64;; esac
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$mode', d['value'], d['ms_per_step'], ' '.join('%.1f'%o['tflops'] for o in d['per_op'][9:]))"
done
