#!/bin/bash
# GPU tool: the regression fuzzers at the current sources, every run under its own timeout; last line of each log into gpurun_out/fuzz/summary.txt
cd "$(dirname "$0")/.."; O=gpurun_out/fuzz; mkdir -p $O
run() { n=$1; shift; echo "== $n: $*" >> $O/summary.txt; timeout 600 env "$@" > $O/$n.log 2>&1; echo "exit $? : $(tail -1 $O/$n.log | cut -c1-600)" >> $O/summary.txt; }
: > $O/summary.txt
run conv_default X=1 python tools/fuzz_conv.py 200 21
run conv_big_default X=1 python tools/fuzz_conv.py 80 26 big
run conv_cbig_force BODAHIP_CBIG=force python tools/fuzz_conv.py 200 22
run conv_cbig_force_big BODAHIP_CBIG=force python tools/fuzz_conv.py 80 27 big
run conv_cbig_split BODAHIP_CBIG=force BODAHIP_CBIG_SPLIT_MIN_GFLOP=0.05 python tools/fuzz_conv.py 80 28 big
run conv_tile_128 X=1 python tools/fuzz_conv.py 80 23 big 128x128x16x2x2x2x1x32x2x2
run conv_tile_64 X=1 python tools/fuzz_conv.py 100 24 small 64x64x16x2x2x2x1x32x2x2
run conv_tile_32x128 X=1 python tools/fuzz_conv.py 100 25 small 32x128x16x1x4x2x1x32x1x2
run nhwc X=1 python tools/fuzz_nhwc.py 300 31
run pool_lrn X=1 python tools/fuzz_pool_lrn.py 40 32
run k1_chain X=1 python tools/fuzz_k1_chain.py 120 33
run lrn_pool_lds X=1 python tools/fuzz_lrn_pool_lds.py 40 34
run sgemm_parts X=1 python tools/fuzz_sgemm_parts.py 60 5
cat $O/summary.txt
