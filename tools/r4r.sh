mkdir -p gpurun_out/s9
for wb in "alexnet 256" "nin 256" "nin 128"; do set -- $wb; echo "== $1 $2"; python tools/tune_tiles.py --workload $1 --batch $2 --iters 20 --min-gain 0.02 2>&1 | tail -16; done | tee gpurun_out/s9/tune.txt
