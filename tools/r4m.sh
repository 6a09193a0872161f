#!/bin/bash
mkdir -p gpurun_out/r4m; O=gpurun_out/r4m
python -m pytest tests/test_gpu_fullnet.py tests/test_gpu_multi.py -m gpu -q -p no:cacheprovider > $O/pytest.log 2>&1; echo rc=$? >> $O/pytest.log; tail -15 $O/pytest.log
net() { local nm=$1; shift
  python bench.py --dtype bf16 --layout nhwc --graph --steps 30 --warmup 5 --no-cpu-baseline "$@" 2>$O/err_$nm.log | tail -1 > $O/$nm.json
  python - <<P >> $O/nets.txt
import json
try:
  d=json.load(open("$O/$nm.json")); print("$nm", d.get("images_per_s"), d["ms_per_step"], d["roofline"].get("conv_ms"), d["roofline"].get("non_conv_ms"), len(d.get("per_call", [])))
except Exception as e: print("$nm FAILED", e)
P
  tail -2 $O/err_$nm.log >> $O/nets.txt
}
net g_default --workload googlenet-net
net g_parallel --workload googlenet-net --parallel-branches
net g_nogrp --workload googlenet-net --no-groups-in-sets
net nin --workload nin-net --batch 256
net alex --workload alexnet-net --batch 256
cat $O/nets.txt
python - <<P
import json
d=json.load(open("$O/g_default.json"))
for c in d["per_call"]: print(f"{c['ms']*1e3:7.1f} {c['func']:22s} {c['tag'][:90]}")
P
