timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -2
python tools/tile_sweep.py --workload alexnet --ops 0 --iters 40 --tiles 96x256x16x1x4x2,96x128x16x1x2x2,96x256x16x1x4x1,96x512x16x1x8x1,32x256x16x1x4x2,96x192x16x1x3x2 2>&1 | grep "^op"
BODAHIP_NO_ROW_GATHER=1 python tools/tile_sweep.py --workload alexnet --ops 0 --iters 40 --tiles 96x256x16x1x4x2,96x128x16x1x2x2 2>&1 | grep "^op"
