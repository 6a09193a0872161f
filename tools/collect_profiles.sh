#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1500 -- 'bash tools/collect_profiles.sh final'
# then, back here:  python tools/make_profile_summaries.py gpurun_out/final r01
# Kernel-trace stats and the PMC passes are separate runs (one --pmc set per run, never combined with other trace domains).
D=${1:-final}
R=$PWD
O=$R/gpurun_out/$D
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SQ="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/calib -o c -- python $R/tools/fetch_calib.py > $O/calib.log 2>&1
for w in sgemm-ops-full alexnet nin; do
  rocprofv3 --kernel-trace --stats -d $O/stats_$w -o p -- python $R/bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline > $O/stats_$w.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch_$w -o p -- python $R/bench.py --workload $w --steps 1 --warmup 0 --settle-ms 0 --no-cpu-baseline > $O/fetch_$w.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write_$w -o p -- python $R/bench.py --workload $w --steps 1 --warmup 0 --settle-ms 0 --no-cpu-baseline > $O/write_$w.log 2>&1
  rocprofv3 --kernel-trace --pmc $SQ -d $O/sq_$w -o p -- python $R/bench.py --workload $w --steps 1 --warmup 0 --settle-ms 0 --no-cpu-baseline > $O/sq_$w.log 2>&1
done
rocprofv3 --kernel-trace --stats -d $O/stats_alexnet_winograd -o p -- python $R/bench.py --workload alexnet --conv-algo winograd --steps 5 --warmup 2 --no-cpu-baseline > $O/stats_alexnet_winograd.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/stats_alexnet_bf16 -o p -- python $R/bench.py --workload alexnet --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline > $O/stats_alexnet_bf16.log 2>&1
cd $R
for w in sgemm-ops-full alexnet nin; do python bench.py --workload $w > $O/bench_$w.json 2>$O/bench_$w.err; done
for w in nin-net alexnet-net googlenet-net googlenet resnet50; do python bench.py --workload $w --no-cpu-baseline > $O/bench_$w.json 2>/dev/null; done
for w in sgemm-ops-full alexnet nin googlenet resnet50; do python bench.py --workload $w --dtype bf16 --no-cpu-baseline > $O/bench_${w}_bf16.json 2>/dev/null; done
for w in alexnet nin googlenet resnet50 alexnet-net; do python bench.py --workload $w --conv-algo winograd --no-cpu-baseline > $O/bench_${w}_winograd.json 2>/dev/null; done
for w in nin-net alexnet-net googlenet-net; do python bench.py --workload $w --dtype bf16 --no-cpu-baseline > $O/bench_${w}_bf16.json 2>/dev/null; done
find $O -name "*.db" -size +30M -delete   # keep the merge-back under the 64 MiB cap
ls -la $O | head -50
for w in googlenet resnet50; do python bench.py --workload $w --batch 256 --no-cpu-baseline > $O/bench_${w}_b256.json 2>/dev/null; python bench.py --workload $w --batch 256 --dtype bf16 --no-cpu-baseline > $O/bench_${w}_b256_bf16.json 2>/dev/null; done
python bench.py --workload googlenet-net --graph --no-cpu-baseline > $O/bench_googlenet-net_graph.json 2>/dev/null
python bench.py --workload googlenet-net --graph --parallel-branches --no-cpu-baseline > $O/bench_googlenet-net_graph_parallel.json 2>/dev/null
for w in googlenet resnet50; do python bench.py --workload $w --graph --no-cpu-baseline > $O/bench_${w}_graph.json 2>/dev/null; done
python bench.py --workload googlenet-net --dtype bf16 --graph --parallel-branches --no-cpu-baseline > $O/bench_googlenet-net_bf16_graph_parallel.json 2>/dev/null
python bench.py --workload resnet50 --conv-algo winograd --graph --no-cpu-baseline > $O/bench_resnet50_winograd_graph.json 2>/dev/null
python bench.py --workload nin-net --batch 128 --no-cpu-baseline > $O/bench_nin-net_b128.json 2>/dev/null
python bench.py --workload nin --batch 128 --no-cpu-baseline > $O/bench_nin_b128.json 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
