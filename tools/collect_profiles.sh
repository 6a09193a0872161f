#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 2400 -- 'bash tools/collect_profiles.sh final'
# then, back here:  python tools/make_profile_summaries.py gpurun_out/final r06
# Kernel-trace stats and the PMC passes are separate runs (one --pmc set per run, never combined with other trace domains).
D=${1:-final}
R=$PWD
O=$R/gpurun_out/$D
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SQ="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
[ -z "$BENCH_ONLY" ] && rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/calib -o c -- python $R/tools/fetch_calib.py > $O/calib.log 2>&1
prof() {   # key, bench args
  k=$1; shift
  rocprofv3 --kernel-trace --stats -d $O/stats_$k -o p -- python $R/bench.py "$@" --steps 20 --warmup 5 --no-cpu-baseline > $O/stats_$k.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch_$k -o p -- python $R/bench.py "$@" --steps 1 --warmup 0 --settle-ms 0 --no-cpu-baseline > $O/fetch_$k.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write_$k -o p -- python $R/bench.py "$@" --steps 1 --warmup 0 --settle-ms 0 --no-cpu-baseline > $O/write_$k.log 2>&1
  rocprofv3 --kernel-trace --pmc $SQ -d $O/sq_$k -o p -- python $R/bench.py "$@" --steps 1 --warmup 0 --settle-ms 0 --no-cpu-baseline > $O/sq_$k.log 2>&1
}
profnet() {   # key, bench args: whole nets -- kernel trace of the timed form, and ONE forward pass under each counter set (BENCH_SINGLE_PASS)
  k=$1; shift
  rocprofv3 --kernel-trace --stats -d $O/stats_$k -o p -- python $R/bench.py "$@" --steps 20 --warmup 5 --no-cpu-baseline > $O/stats_$k.log 2>&1
  BENCH_SINGLE_PASS=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch_$k -o p -- python $R/bench.py "$@" --no-cpu-baseline > $O/fetch_$k.log 2>&1
  BENCH_SINGLE_PASS=1 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write_$k -o p -- python $R/bench.py "$@" --no-cpu-baseline > $O/write_$k.log 2>&1
  BENCH_SINGLE_PASS=1 rocprofv3 --kernel-trace --pmc $SQ -d $O/sq_$k -o p -- python $R/bench.py "$@" --no-cpu-baseline > $O/sq_$k.log 2>&1
  echo 1 > $O/passes_$k.txt
}
if [ -z "$BENCH_ONLY" ]; then    # BENCH_ONLY=1: only the bench lines (second pass, once profiles/pmc_summary.json carries the current kernel-source hash: roofline.traffic is then filled)
prof sgemm-ops-full --workload sgemm-ops-full --no-conv-ops
prof alexnet --workload alexnet
prof nin --workload nin
prof googlenet-bf16-nhwc --workload googlenet --dtype bf16 --layout nhwc
prof resnet50-bf16-nhwc --workload resnet50 --dtype bf16 --layout nhwc
profnet googlenet-net-bf16-nhwc --workload googlenet-net --dtype bf16 --layout nhwc
profnet nin-net-b128 --workload nin-net --batch 128
fi
cd $R
python -c "import bench; print(bench.kernel_src_hash())" > $O/kernel_src_hash.txt
python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2>$O/bench_default.err      # the driver's command: headline + conv_ops + configs legs + cpu baselines + compile
for w in alexnet nin; do python bench.py --workload $w > $O/bench_$w.json 2>$O/bench_$w.err; done
python bench.py --workload alexnet --exact 0 --no-cpu-baseline > $O/bench_alexnet_tolerance.json 2>/dev/null
B="--dtype bf16 --layout nhwc --steps 20 --warmup 5 --no-cpu-baseline"
for w in googlenet resnet50; do
  python bench.py --workload $w $B --graph > $O/bench_${w}_bf16_nhwc_graph.json 2>/dev/null
  python bench.py --workload $w $B --graph --independent > $O/bench_${w}_bf16_nhwc_graph_independent.json 2>/dev/null
  python bench.py --workload $w $B --graph --independent --multi > $O/bench_${w}_bf16_nhwc_graph_independent_multi.json 2>/dev/null
  # round 5 A/B on this box: the two-kernel form of the K slices (slabs in the shared scratch + a reduce pass) and no slices at all
  BODAHIP_NHWC_SPLITK2=1 python bench.py --workload $w $B --graph --independent > $O/bench_${w}_bf16_nhwc_graph_independent_splitk2.json 2>/dev/null
  BODAHIP_NO_NHWC_SPLITK=1 python bench.py --workload $w $B --graph --independent > $O/bench_${w}_bf16_nhwc_graph_independent_nosplitk.json 2>/dev/null
done
for w in nin-net alexnet-net googlenet-net; do
  python bench.py --workload $w $B --graph > $O/bench_${w}_bf16_nhwc_graph.json 2>/dev/null
done
python bench.py --workload googlenet-net $B --graph --no-fuse-pools > $O/bench_googlenet-net_bf16_nhwc_graph_nopoolfusion.json 2>/dev/null
BODAHIP_NHWC_POOL_R4PLAN=1 python bench.py --workload googlenet-net $B --graph > $O/bench_googlenet-net_bf16_nhwc_graph_r4poolplan.json 2>/dev/null      # round 5 A/B: the round-4 tiles of the fused-pooling convolutions
BODAHIP_NO_NHWC_SPLITK=1 python bench.py --workload googlenet-net $B --graph > $O/bench_googlenet-net_bf16_nhwc_graph_nosplitk.json 2>/dev/null          # ... and no K slices anywhere
python bench.py --workload nin-net --batch 128 --graph --no-cpu-baseline > $O/bench_nin-net_b128_graph.json 2>/dev/null
python bench.py --workload nin-net --batch 128 --exact 0 --graph --no-cpu-baseline > $O/bench_nin-net_b128_graph_tolerance.json 2>/dev/null
python bench.py --workload nin --batch 128 --no-cpu-baseline > $O/bench_nin_b128.json 2>/dev/null
find $O -name "*.db" -size +30M -delete   # keep the merge-back under the 64 MiB cap
ls $O | head -80
