#!/bin/bash
# GPU tool: 5120^3 / 6144^3 / 7168^3 of sgemm-ops-full (layer sequence, TF/s) as ONE launch under each tile form of sgemm_big_f32.hip, against the two-level split (base)
cd "$(dirname "$0")/.."
run() { BODAHIP_SGEMM_TILE_FOR="$2" python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-conv-ops 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('%-12s' % '$1', 'total %.2f' % d['value'], ' '.join('%.1f' % o['tflops'] for o in d['per_op'][10:14]))
"; }
echo "sizes: 4096 5120 6144 7168"
run base ""
run w256x128 "5120=256x128x8x3x4x1;6144=256x128x8x3x4x1;7168=256x128x8x3x4x1"
run w128x256 "5120=128x256x8x3x4x1;6144=128x256x8x3x4x1;7168=128x256x8x3x4x1"
run w256x256 "5120=256x256x8x3x4x1;6144=256x256x8x3x4x1;7168=256x256x8x3x4x1"
run w128x128 "5120=128x128x8x3x4x2;6144=128x128x8x3x4x2;7168=128x128x8x3x4x2"
