for ab in 0 14 4 10; do echo "ABLATE=$ab"; BODAHIP_EXTRA_DEFS="-DABLATE=$ab" python tools/tile_sweep.py --workload alexnet --ops 1,3 --iters 15 2>&1 | grep auto; done
