#!/bin/bash
# GPU tool: channel groups per K step of the fused-pooling convolutions (BODAHIP_NHWC_POOL_CG) x tile (BODAHIP_NHWC_POOL_TILE); per-call us of GoogLeNet's icpN_out3 at 64 images
cd "$(dirname "$0")/.."
for T in auto 64x128x0x1x4x2 64x64x0x1x2x2; do
for CG in 4 8 16 32; do
  if [ "$T" = auto ]; then unset BODAHIP_NHWC_POOL_TILE; else export BODAHIP_NHWC_POOL_TILE=$T; fi
  export BODAHIP_NHWC_POOL_CG=$CG
  python bench.py --workload googlenet-net --dtype bf16 --layout nhwc --no-fuse-levels --steps 5 --warmup 2 --no-cpu-baseline 2>/tmp/err.txt | python -c "
import sys, json
ls = [l for l in sys.stdin if l.startswith('{')]
if not ls: print('$T cg$CG: failed', open('/tmp/err.txt').read()[-300:]); sys.exit(0)
d = json.loads(ls[0])
pc = {c['tag']: c['ms'] for c in d['per_call']}
o3 = [t for t in pc if t.endswith('_out3')]
print('%-18s cg%-3s' % ('$T', '$CG'), ' '.join('%5.1f' % (pc[t] * 1e3) for t in o3), ' | sum %.1f us | %.1f k img/s' % (sum(pc[t] for t in o3) * 1e3, d['images_per_s'] / 1e3))
"
done; done
