mkdir -p gpurun_out/s6
python -m pytest tests/test_gpu_parity.py -x -q -k "sgemm" 2>&1 | tail -2
for d in "-DNSTG=4" "-DNSTG=3"; do for v in 16x2 8x4 8x2 16x4; do echo "$d SGEMM_BIG=$v"; BODAHIP_EXTRA_DEFS="$d" BODAHIP_SGEMM_BIG=$v python tools/tile_sweep.py --workload sgemm --ops 10,12,14,16 --iters 6 2>&1 | grep auto; done; done | tee gpurun_out/s6/sgemm_big2.txt
