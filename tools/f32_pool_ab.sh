#!/bin/bash
# fp32 max pooling fused into the consuming convolution (round 5): parity tests, then same-box A/B of the fp32 nets with the fusion on / off.
#   gpurun --timeout 1500 -- 'bash tools/f32_pool_ab.sh'
O=gpurun_out/f32pool; mkdir -p $O
timeout 1100 python -m pytest tests/test_gpu_fullnet.py "tests/test_gpu_parity.py::test_conv_with_a_max_pooling_fused_in_front_bit_exact" -q -m gpu > $O/tests.log 2>&1; echo "tests exit $?" >> $O/tests.log
for rep in 1 2; do
for net in "nin-net 128" "alexnet-net 128"; do
  set -- $net
  for mode in on off; do
    if [ $mode = on ]; then export BODAHIP_F32_POOL_FUSION=1; else unset BODAHIP_F32_POOL_FUSION; fi
    timeout 300 python bench.py --workload $1 --batch $2 --graph --no-cpu-baseline --steps 30 --warmup 5 > $O/${1}_${mode}_$rep.json 2> $O/${1}_${mode}_$rep.err
  done
done
done
export BODAHIP_F32_POOL_FUSION=1
BODAHIP_F32_POOL_ALL=1 timeout 300 python bench.py --workload nin-net --batch 128 --graph --no-cpu-baseline --steps 30 --warmup 5 > $O/nin-net_all.json 2> $O/nin-net_all.err
python - <<'P'
import json, glob
for f in sorted(glob.glob("gpurun_out/f32pool/*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], j["value"], j["unit"], j["ms_per_step"], j.get("roofline", {}).get("frac"))
    except Exception as e:
        print(f, "ERR", e)
P
tail -5 $O/tests.log
