mkdir -p gpurun_out/s5
python -m pytest tests/test_gpu_parity.py -x -q -k "fc_kernel" 2>&1 | tail -3
for fc in off "" 32x32x64x2 32x32x64x4 32x32x32x4 32x64x64x2 64x32x64x2; do echo "FC=$fc"; BODAHIP_FC=$fc python tools/tile_sweep.py --workload alexnet --ops 5,6,7 --iters 30 2>&1 | grep auto; done
for w in googlenet resnet50; do python tools/tile_sweep.py --workload $w --batch 64 --ops $( [ $w = googlenet ] && echo 63 || echo 53 ) --iters 30 2>&1 | grep auto; BODAHIP_FC=off python tools/tile_sweep.py --workload $w --batch 64 --ops $( [ $w = googlenet ] && echo 63 || echo 53 ) --iters 30 2>&1 | grep auto; done
