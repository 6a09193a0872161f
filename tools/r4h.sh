#!/bin/bash
mkdir -p gpurun_out/r4h; O=gpurun_out/r4h
python -m pytest tests/test_gpu_nhwc.py tests/test_gpu_fullnet.py -m gpu -q -p no:cacheprovider -x -k "level_set or sibling_fusion or multi_problem" > $O/pytest.log 2>&1; echo rc=$? >> $O/pytest.log; tail -30 $O/pytest.log
net() { local nm=$1; shift
  python bench.py --workload googlenet-net --dtype bf16 --layout nhwc --graph --parallel-branches --steps 30 --warmup 5 --no-cpu-baseline "$@" 2>$O/err_$nm.log | tail -1 > $O/$nm.json
  python - <<P >> $O/nets.txt
import json
try:
  d=json.load(open("$O/$nm.json")); print("$nm", d["images_per_s"], d["ms_per_step"], "conv_ms", d["roofline"]["conv_ms"], "non_conv", d["roofline"]["non_conv_ms"], "calls", len(d["per_call"]))
except Exception as e: print("$nm FAILED", e)
P
  tail -3 $O/err_$nm.log >> $O/nets.txt
}
net sets
net nosets --no-fuse-levels
net sets_chain_graph
python bench.py --workload googlenet-net --dtype bf16 --layout nhwc --graph --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sets chain', d['images_per_s'], d['ms_per_step'])" >> $O/nets.txt
python bench.py --workload googlenet-net --dtype bf16 --layout nhwc --graph --no-fuse-levels --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('nosets chain', d['images_per_s'], d['ms_per_step'])" >> $O/nets.txt
cat $O/nets.txt
python - <<P
import json
d=json.load(open("$O/sets.json"))
for c in d["per_call"]: print(f"{c['ms']*1e3:7.1f} {c['func']:22s} {c['tag'][:60]}")
P
