#!/bin/bash
# Rolling-rows kernel (round 5): parity tests, then same-box A/Bs -- GoogLeNet net @64 with conv1 + pool1 + norm1 as one launch / apart / the patch kernel; the config-5 lists.
#   gpurun --timeout 2400 -- 'bash tools/rows_ab.sh'
O=gpurun_out/rows; mkdir -p $O
timeout 1500 python -m pytest "tests/test_gpu_nhwc.py::test_rolling_rows_kernel_vs_oracle_and_patch_kernel" "tests/test_gpu_fullnet.py::test_pooling_and_lrn_taken_into_the_convolutions_launch_are_bit_identical" -q -m gpu -x > $O/tests.log 2>&1; echo "tests exit $?" >> $O/tests.log
tail -25 $O/tests.log
G="--workload googlenet-net --dtype bf16 --layout nhwc --graph --no-cpu-baseline --steps 30 --warmup 5"
for rep in 1 2; do
  timeout 300 python bench.py $G > $O/gnet_fused_$rep.json 2> $O/gnet_fused_$rep.err
  timeout 300 python bench.py $G --no-fuse-post > $O/gnet_rowsapart_$rep.json 2> $O/gnet_rowsapart_$rep.err
  BODAHIP_NHWC_ROWS=0 timeout 300 python bench.py $G --no-fuse-post > $O/gnet_patch_$rep.json 2> $O/gnet_patch_$rep.err
done
timeout 300 python bench.py --workload googlenet-net --dtype bf16 --layout nhwc --per-op --timing kernel --no-cpu-baseline --steps 20 --warmup 5 > $O/perop.json 2> $O/perop.err
L="--dtype bf16 --layout nhwc --graph --independent --no-cpu-baseline --steps 30 --warmup 5"
for net in googlenet resnet50; do
  timeout 300 python bench.py --workload $net $L > $O/list_${net}_rows.json 2> $O/list_${net}_rows.err
  BODAHIP_NHWC_ROWS=0 timeout 300 python bench.py --workload $net $L > $O/list_${net}_patch.json 2> $O/list_${net}_patch.err
done
python - <<'P'
import json, glob
for f in sorted(glob.glob("gpurun_out/rows/*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], j["value"], j["unit"], j["ms_per_step"], j.get("images_per_s"), j.get("roofline", {}).get("frac"))
    except Exception as e:
        print(f, "ERR", e)
P
grep -v "^\[" $O/perop.err | head -8
