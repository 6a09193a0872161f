mkdir -p gpurun_out/s8
python tools/tile_sweep.py --workload alexnet --ops 0 --iters 20 --tiles 96x256x22x1x4x2x1x32x1x0,96x256x22x1x4x2x1x32x1x1,96x256x22x1x4x1x1x32x1x1,96x128x22x1x2x2x1x32x1x1,96x128x22x1x2x4x1x32x1x1 2>&1 | tee gpurun_out/s8/specw.txt
python tools/tile_sweep.py --workload alexnet --ops 1,2,4 --iters 20 --tiles 64x256x36x1x4x2x1x32x1x1,64x256x36x1x4x1x1x32x1x1,128x256x36x2x4x1x1x32x1x1,32x256x36x1x4x2x1x32x1x1 2>&1 | tee -a gpurun_out/s8/specw.txt
python tools/tile_sweep.py --workload nin --ops 4,7,10 --iters 20 --tiles 128x128x16x2x2x2x1x32x1x1,96x256x16x1x4x2x1x32x1x1,64x64x16x2x2x2x1x32x1x1,128x128x32x2x2x2x1x32x1x1 2>&1 | tee -a gpurun_out/s8/specw.txt
