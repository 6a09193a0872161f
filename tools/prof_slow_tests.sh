#!/bin/bash
# where do the slow GPU tests spend their time?  (cProfile of two GoogLeNet tests + the code objects compiled at run time)
mkdir -p gpurun_out/r4b; export BODAHIP_CACHE_LOG=$PWD/gpurun_out/r4b/cache_miss.log; rm -f $BODAHIP_CACHE_LOG
python -m cProfile -o gpurun_out/r4b/p1.prof -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -k "test_config5_every_layer_bf16_at_bench_batch and googlenet" > gpurun_out/r4b/p1.log 2>&1
python -m cProfile -o gpurun_out/r4b/p2.prof -m pytest tests/test_gpu_fullnet.py -q -p no:cacheprovider -k "test_full_net_forward_bf16_operands and googlenet" > gpurun_out/r4b/p2.log 2>&1
python - <<'P' > gpurun_out/r4b/prof.txt 2>&1
import pstats
for f in ("gpurun_out/r4b/p1.prof", "gpurun_out/r4b/p2.prof"):
    print("=====", f)
    pstats.Stats(f).sort_stats("cumulative").print_stats(45)
    pstats.Stats(f).sort_stats("tottime").print_stats(25)
P
tail -3 gpurun_out/r4b/p1.log gpurun_out/r4b/p2.log; wc -l $BODAHIP_CACHE_LOG
