python tools/tile_sweep.py --workload alexnet --ops 1 --iters 20 --tiles 32x256x50x1x4x2,32x256x50x1x4x3,32x256x50x1x4x4,64x256x50x1x4x2,64x256x50x1x4x3,32x128x50x1x4x4,32x128x50x1x4x6,32x256x50x1x4x2x1x32x2 2>&1
python tools/tile_sweep.py --workload alexnet --ops 3 --iters 20 --tiles 64x256x36x1x4x2,64x256x36x1x4x3,32x256x36x1x4x3,32x256x36x1x4x4 2>&1
