#!/bin/bash
mkdir -p gpurun_out/r4l; O=gpurun_out/r4l
python -m pytest tests/test_gpu_nhwc.py tests/test_gpu_fullnet.py tests/test_gpu_multi.py -m gpu -q -p no:cacheprovider > $O/pytest.log 2>&1; echo rc=$? >> $O/pytest.log; tail -15 $O/pytest.log
net() { local nm=$1; shift
  python bench.py --dtype bf16 --layout nhwc --graph --steps 30 --warmup 5 --no-cpu-baseline "$@" 2>$O/err_$nm.log | tail -1 > $O/$nm.json
  python - <<P >> $O/nets.txt
import json
try:
  d=json.load(open("$O/$nm.json")); print("$nm", d.get("images_per_s"), d["ms_per_step"], d["roofline"].get("conv_ms"), d["roofline"].get("non_conv_ms"), d["roofline"].get("timed_region"), len(d.get("per_call", [])), d["config"].get("level_sets"))
except Exception as e: print("$nm FAILED", e)
P
  tail -2 $O/err_$nm.log >> $O/nets.txt
}
net g_sets --workload googlenet-net
net g_pools --workload googlenet-net --fuse-pools
net l_goog_sets8 --workload googlenet --independent --sets 8
net l_goog_sets16 --workload googlenet --independent --sets 16
net l_goog_multi_sets8 --workload googlenet --independent --multi --sets 8
net l_goog_sets8_chain --workload googlenet --sets 8
net l_res_sets8 --workload resnet50 --independent --sets 8
net l_res_sets16 --workload resnet50 --independent --sets 16
net l_res_sets4 --workload resnet50 --independent --sets 4
cat $O/nets.txt
