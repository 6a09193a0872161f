#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 2400 -- 'bash tools/collect_profiles.sh final'
# then, back here:  python tools/make_profile_summaries.py gpurun_out/final r04
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
  rocprofv3 --kernel-trace --stats -d $O/stats_$k -o p -- python $R/bench.py "$@" --steps 5 --warmup 2 --no-cpu-baseline > $O/stats_$k.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch_$k -o p -- python $R/bench.py "$@" --steps 1 --warmup 0 --settle-ms 0 --no-cpu-baseline > $O/fetch_$k.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write_$k -o p -- python $R/bench.py "$@" --steps 1 --warmup 0 --settle-ms 0 --no-cpu-baseline > $O/write_$k.log 2>&1
  rocprofv3 --kernel-trace --pmc $SQ -d $O/sq_$k -o p -- python $R/bench.py "$@" --steps 1 --warmup 0 --settle-ms 0 --no-cpu-baseline > $O/sq_$k.log 2>&1
}
if [ -z "$BENCH_ONLY" ]; then    # BENCH_ONLY=1: only the bench lines (second pass, once profiles/pmc_summary.json carries the current kernel-source hash: roofline.traffic is then filled)
prof sgemm-ops-full --workload sgemm-ops-full --no-conv-ops
prof alexnet --workload alexnet
prof nin --workload nin
prof googlenet-bf16-nhwc --workload googlenet --dtype bf16 --layout nhwc
prof resnet50-bf16-nhwc --workload resnet50 --dtype bf16 --layout nhwc
fi
cd $R
python -c "import bench; print(bench.kernel_src_hash())" > $O/kernel_src_hash.txt
python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2>$O/bench_default.err      # the driver's command: headline + conv_ops + configs legs
for w in alexnet nin; do python bench.py --workload $w > $O/bench_$w.json 2>$O/bench_$w.err; done
for w in googlenet resnet50; do
  python bench.py --workload $w --no-cpu-baseline > $O/bench_$w.json 2>/dev/null
  python bench.py --workload $w --dtype bf16 --layout nhwc --no-cpu-baseline > $O/bench_${w}_bf16_nhwc.json 2>/dev/null
  python bench.py --workload $w --dtype bf16 --layout nhwc --graph --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_${w}_bf16_nhwc_graph.json 2>/dev/null
  python bench.py --workload $w --dtype bf16 --layout nhwc --graph --steps 20 --warmup 5 --no-patch --no-cpu-baseline > $O/bench_${w}_bf16_nhwc_graph_nopatch.json 2>/dev/null
done
python bench.py --workload googlenet --dtype bf16 --layout nhwc --graph --steps 20 --warmup 5 --group-siblings --no-cpu-baseline > $O/bench_googlenet_bf16_nhwc_graph_grouped.json 2>/dev/null
# round 4: the lists with the graph's edges removed, and with their implicit-GEMM members as ONE multi-problem launch
for w in googlenet resnet50; do
  python bench.py --workload $w --dtype bf16 --layout nhwc --graph --independent --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_${w}_bf16_nhwc_graph_independent.json 2>/dev/null
  python bench.py --workload $w --dtype bf16 --layout nhwc --graph --multi --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_${w}_bf16_nhwc_graph_multi.json 2>/dev/null
  python bench.py --workload $w --dtype bf16 --layout nhwc --graph --independent --multi --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_${w}_bf16_nhwc_graph_independent_multi.json 2>/dev/null
done
python bench.py --workload googlenet --dtype bf16 --layout nhwc --graph --independent --multi --sets 8 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_googlenet_bf16_nhwc_graph_independent_multi_sets8.json 2>/dev/null
for w in nin-net alexnet-net googlenet-net; do
  python bench.py --workload $w --no-cpu-baseline --graph --parallel-branches > $O/bench_${w}_graph.json 2>/dev/null
  python bench.py --workload $w --dtype bf16 --layout nhwc --no-cpu-baseline --graph --steps 20 --warmup 5 > $O/bench_${w}_bf16_nhwc_graph.json 2>/dev/null      # round 4 default: level sets, groups in sets, fused poolings; chain graph
  python bench.py --workload $w --dtype bf16 --layout nhwc --no-cpu-baseline --graph --parallel-branches --steps 20 --warmup 5 --no-fuse-levels --no-fuse-pools > $O/bench_${w}_bf16_nhwc_graph_r03form.json 2>/dev/null   # the round-3 form on this box
done
python bench.py --workload googlenet-net --dtype bf16 --layout nhwc --no-cpu-baseline --graph --steps 20 --warmup 5 --no-fuse-pools > $O/bench_googlenet-net_bf16_nhwc_graph_nopoolfusion.json 2>/dev/null
for w in googlenet-net alexnet-net; do python bench.py --workload $w --dtype bf16 --layout nhwc --no-cpu-baseline --graph --steps 20 --warmup 5 --no-fuse-pool-lrn > $O/bench_${w}_bf16_nhwc_graph_nopoollrnfusion.json 2>/dev/null; done   # round 4b: pooling + LRN as two passes
for w in googlenet-net alexnet-net; do BENCH_FUSE_POOL_LRN=all python bench.py --workload $w --dtype bf16 --layout nhwc --no-cpu-baseline --graph --steps 20 --warmup 5 > $O/bench_${w}_bf16_nhwc_graph_lrnfirstfused.json 2>/dev/null; done   # round 4c: LRN -> Pooling pairs fused too (the default fuses Pooling -> LRN pairs only)
for w in googlenet-net alexnet-net; do BODAHIP_NO_LRN_POOL_LDS=1 python bench.py --workload $w --dtype bf16 --layout nhwc --no-cpu-baseline --graph --steps 20 --warmup 5 > $O/bench_${w}_bf16_nhwc_graph_nolrnpoollds.json 2>/dev/null; done   # round 4c: LRN -> Pooling pairs apart (default: through LDS)
python bench.py --workload googlenet-net --dtype bf16 --layout nhwc --no-cpu-baseline --graph --steps 20 --warmup 5 --no-groups-in-sets > $O/bench_googlenet-net_bf16_nhwc_graph_nogroupsinsets.json 2>/dev/null
cd /tmp; rocprofv3 --kernel-trace --stats -d $O/stats_googlenet-net-bf16-nhwc -o p -- python $R/bench.py --workload googlenet-net --dtype bf16 --layout nhwc --steps 5 --warmup 2 --no-cpu-baseline > $O/stats_googlenet-net.log 2>&1; cd $R
# round 4c: the fp32 NiN net at config 4's per-GPU batch (hip_conv_k1_chain inside): kernel trace, and the SQ counters per kernel
cd /tmp; rocprofv3 --kernel-trace --stats -d $O/stats_nin-net-b128 -o p -- python $R/bench.py --workload nin-net --batch 128 --steps 5 --warmup 2 --no-cpu-baseline > $O/stats_nin-net.log 2>&1
rocprofv3 --kernel-trace --pmc $SQ -d $O/sq_nin-net-b128 -o p -- python $R/bench.py --workload nin-net --batch 128 --steps 1 --warmup 0 --settle-ms 0 --no-cpu-baseline > $O/sq_nin-net.log 2>&1; cd $R
python bench.py --workload alexnet --exact 0 --no-cpu-baseline > $O/bench_alexnet_tolerance.json 2>/dev/null
python bench.py --workload nin-net --batch 128 --no-cpu-baseline > $O/bench_nin-net_b128.json 2>/dev/null
python bench.py --workload nin-net --batch 128 --exact 0 --no-cpu-baseline > $O/bench_nin-net_b128_tolerance.json 2>/dev/null
python bench.py --workload nin --batch 128 --no-cpu-baseline > $O/bench_nin_b128.json 2>/dev/null
python bench.py --workload nin-net --batch 128 --graph --no-cpu-baseline > $O/bench_nin-net_b128_graph.json 2>/dev/null    # round 4c: config 4's leg of the default line (one hipGraph replay per forward pass)
python bench.py --workload nin-net --batch 128 --no-fuse-k1-chains --no-cpu-baseline > $O/bench_nin-net_b128_nochain.json 2>/dev/null    # round 4c: cccp1 / cccp2 as two launches (default: one hip_conv_k1_chain call)
find $O -name "*.db" -size +30M -delete   # keep the merge-back under the 64 MiB cap
ls $O | head -80
