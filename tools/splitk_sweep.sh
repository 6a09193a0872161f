T="64x64x32x2x2x2x2x32x2,64x64x32x2x2x2x4x32x2,64x64x32x2x2x2x8x32x2,64x64x32x2x2x2x4x32x1,128x128x16x2x2x2x4,128x128x16x2x2x2x8,128x128x16x2x2x2x16,128x128x32x2x2x2x8,64x64x16x2x2x2x4x32x2"
python tools/tile_sweep.py --workload alexnet --ops 5,6,7 --iters 12 --tiles $T
