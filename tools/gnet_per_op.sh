#!/bin/bash
# GoogLeNet net @64 (bf16 channels-last): the step's per-call table (call by call, backend events) beside the graph-replay number.
O=gpurun_out/gnet_per_op; mkdir -p $O
timeout 300 python bench.py --workload googlenet-net --dtype bf16 --layout nhwc --graph --no-cpu-baseline --steps 30 --warmup 5 > $O/graph.json 2> $O/graph.err
timeout 300 python bench.py --workload googlenet-net --dtype bf16 --layout nhwc --per-op --timing kernel --no-cpu-baseline --steps 20 --warmup 5 > $O/perop.json 2> $O/perop.err
tail -1 $O/graph.json | cut -c1-400; grep -v "^\[" $O/perop.err | head -120
