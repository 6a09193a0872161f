#!/bin/bash
# GPU tool: per-size TF/s of sgemm-ops-full in the layer sequence (warm clocks) under per-size tile overrides (BODAHIP_SGEMM_TILE_FOR); one line per variant, alternating
cd "$(dirname "$0")/.."
run() { BODAHIP_SGEMM_TILE_FOR="$2" python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-conv-ops 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('%-10s' % '$1', 'total %.2f TF/s' % d['value'], ' '.join('%.1f' % o['tflops'] for o in d['per_op'][6:]))
"; }
echo "sizes: 1024 1536 2048 3072 4096 5120 6144 7168 8192 10240 12288"
D="2048=128x128x8x3x4x2;6144=128x128x16x3x4x2;4096=256x128x8x3x4x1;8192=256x128x8x3x4x1;10240=256x128x8x3x4x1;12288=256x128x8x3x4x1"
E2="2048=128x128x8x3x4x2;4096=128x256x8x3x4x1;8192=128x256x8x3x4x1;10240=128x256x8x3x4x1;12288=128x256x8x3x4x1"
for i in 1 2 3; do run base ""; run D "$D"; run E "$E2"; done
