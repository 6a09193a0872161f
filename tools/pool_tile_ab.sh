#!/bin/bash
# GPU tool: the fused-pooling convolutions (an inception module's pool -> pool projection, conv_nhwc_patch_bf16.hip -DPOOL=1) of GoogLeNet at 64 images, each a launch
# of its own (--no-fuse-levels), under a list of tiles (BODAHIP_NHWC_POOL_TILE); per-call ms of the icpN_out3 calls.   tools/pool_tile_ab.sh [tile ...]
cd "$(dirname "$0")/.."
TILES=("$@"); [ ${#TILES[@]} -eq 0 ] && TILES=(auto 128x128x0x1x4x2 64x128x0x1x4x2 64x128x0x2x2x2 128x64x0x2x2x2 64x64x0x1x2x2 128x128x0x2x2x2)
for T in "${TILES[@]}"; do
  if [ "$T" = auto ]; then unset BODAHIP_NHWC_POOL_TILE; else export BODAHIP_NHWC_POOL_TILE=$T; fi
  python bench.py --workload googlenet-net --dtype bf16 --layout nhwc --no-fuse-levels --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0])
pc = {c['tag']: c['ms'] for c in d['per_call']}
o3 = [t for t in pc if t.endswith('_out3')]
print('%-22s' % '$T', ' '.join('%5.1f' % (pc[t] * 1e3) for t in o3), ' | sum %.1f us | %.1f k img/s' % (sum(pc[t] for t in o3) * 1e3, d['images_per_s'] / 1e3))
"
done
