#!/bin/bash
# pooling <-> LRN fusion A/B on the channels-last full nets (one box): python bench.py --workload X-net --dtype bf16 --layout nhwc --graph [--no-fuse-pool-lrn]
for w in googlenet-net alexnet-net; do for f in "" "--no-fuse-pool-lrn"; do
  python bench.py --workload $w --dtype bf16 --layout nhwc --graph --steps 20 --warmup 5 --no-cpu-baseline $f 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', '$f' or '(fused)', d['images_per_s'], d['ms_per_step'], d['roofline'].get('non_conv_ms'))"
done; done
