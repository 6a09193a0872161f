#!/bin/bash
mkdir -p gpurun_out/r4k; O=gpurun_out/r4k
python -m pytest tests/test_gpu_nhwc.py tests/test_gpu_fullnet.py -m gpu -q -p no:cacheprovider -x > $O/pytest.log 2>&1; echo rc=$? >> $O/pytest.log; tail -12 $O/pytest.log
REPS=25 python tools/multi_stress.py > $O/stress.txt 2>&1; cat $O/stress.txt | cut -c1-200
net() { local nm=$1; shift
  python bench.py --dtype bf16 --layout nhwc --graph --steps 30 --warmup 5 --no-cpu-baseline "$@" 2>$O/err_$nm.log | tail -1 > $O/$nm.json
  python - <<P >> $O/nets.txt
import json
try:
  d=json.load(open("$O/$nm.json")); print("$nm", d.get("images_per_s"), d["ms_per_step"], d["roofline"].get("conv_ms"), d["roofline"].get("non_conv_ms"), d["roofline"].get("timed_region"), len(d.get("per_call", [])))
except Exception as e: print("$nm FAILED", e)
P
  tail -2 $O/err_$nm.log >> $O/nets.txt
}
net g_all --workload googlenet-net
net g_nopools --workload googlenet-net --no-fuse-pools
net l_goog_ind --workload googlenet --independent
net l_res_ind --workload resnet50 --independent
net l_goog_chain --workload googlenet
cat $O/nets.txt
python - <<P
import json
d=json.load(open("$O/g_all.json"))
for c in d["per_call"]: print(f"{c['ms']*1e3:7.1f} {c['func']:22s} {c['tag'][:70]}")
P
