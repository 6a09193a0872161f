#!/bin/bash
# GPU tool: the config-5 legs of the default line, one by one (JSON lines into gpurun_out/r5_cfg5_<tag>.json)
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
TAG=${1:-run}
B="--dtype bf16 --layout nhwc --steps 20 --warmup 5 --no-cpu-baseline"
python bench.py --workload googlenet-net $B --graph > gpurun_out/r5_cfg5_${TAG}_googlenet-net.json 2>gpurun_out/err.txt || tail -5 gpurun_out/err.txt
python bench.py --workload googlenet $B --graph --independent > gpurun_out/r5_cfg5_${TAG}_googlenet_indep.json 2>gpurun_out/err.txt || tail -5 gpurun_out/err.txt
python bench.py --workload resnet50 $B --graph --independent > gpurun_out/r5_cfg5_${TAG}_resnet50_indep.json 2>gpurun_out/err.txt || tail -5 gpurun_out/err.txt
python bench.py --workload googlenet $B --graph > gpurun_out/r5_cfg5_${TAG}_googlenet_chain.json 2>gpurun_out/err.txt || tail -5 gpurun_out/err.txt
python - <<P
import json,glob
for fn in sorted(glob.glob("gpurun_out/r5_cfg5_${TAG}_*.json")):
    try:
        d=json.loads([l for l in open(fn) if l.startswith("{")][0])
        r=d["roofline"]
        print(fn.split("${TAG}_")[1], "value", d["value"], "ms/step", d["ms_per_step"], "img/s", d.get("images_per_s"), "frac", r.get("frac"), "wall", (r.get("timed_region") or {}).get("frac"), "conv_ms", r.get("conv_ms"), "non_conv", r.get("non_conv_ms"))
    except Exception as e: print(fn, "ERR", e)
P
