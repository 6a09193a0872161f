#!/bin/bash
# sum of per-layer times under the planner's choices for the five conv workloads (tools/tune_tiles.py without candidates): planner A/B tests
for wb in "nin 128" "nin 256" "alexnet 256" "googlenet 64" "resnet50 64"; do set -- $wb; python tools/tune_tiles.py --workload $1 --batch $2 --tiles "" --iters 20 2>&1 | tail -1; done
