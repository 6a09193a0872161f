#!/bin/bash
mkdir -p gpurun_out/r4i; O=gpurun_out/r4i
python -m pytest tests/test_gpu_nhwc.py tests/test_gpu_fullnet.py -m gpu -q -p no:cacheprovider -x > $O/pytest.log 2>&1; echo rc=$? >> $O/pytest.log; tail -12 $O/pytest.log
net() { local nm=$1; shift
  python bench.py --dtype bf16 --layout nhwc --graph --steps 30 --warmup 5 --no-cpu-baseline "$@" 2>$O/err_$nm.log | tail -1 > $O/$nm.json
  python - <<P >> $O/nets.txt
import json
try:
  d=json.load(open("$O/$nm.json")); print("$nm", d["images_per_s"], d["ms_per_step"], "conv_ms", d["roofline"]["conv_ms"], "non_conv", d["roofline"]["non_conv_ms"], "calls", len(d["per_call"]), "sets", [len(x) for x in d["config"]["level_sets"]])
except Exception as e: print("$nm FAILED", e)
P
  tail -2 $O/err_$nm.log >> $O/nets.txt
}
net g_sets --workload googlenet-net --parallel-branches
net g_sets_chain --workload googlenet-net
net g_sets_nosib --workload googlenet-net --no-fuse-siblings
net g_nosets --workload googlenet-net --no-fuse-levels
net nin --workload nin-net --batch 256
net alex --workload alexnet-net --batch 256
cat $O/nets.txt
