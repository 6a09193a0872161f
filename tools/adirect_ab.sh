# GPU tool: same-box A/B of the patch kernel's two operand paths (BODAHIP_NHWC_ADIRECT=0: both operands staged through the LDS; 1: filter fragments from global memory)
python -m pytest tests/test_gpu_nhwc.py -x -q 2>&1 | tail -3
for ad in 0 1; do export BODAHIP_NHWC_ADIRECT=$ad BODAHIP_CACHE_DIR=/tmp/kc_adab$ad
for w in resnet50 googlenet; do
  python bench.py --workload $w --dtype bf16 --layout nhwc --no-cpu-baseline --steps 20 --warmup 5 --graph 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('direct=$ad $w',d['value'],d['ms_per_step'],r['kernel_ms_per_step'],r['frac'],{k:(v['ops'],v['ms'],v['frac']) for k,v in r['per_bound'].items()}); print('   ',[round(p['ms']*1e3,1) for p in d['per_op']])"
done
python bench.py --workload googlenet --dtype bf16 --layout nhwc --no-cpu-baseline --steps 20 --warmup 5 --graph --group-siblings 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('direct=$ad googlenet grouped',d['value'],d['ms_per_step'],r['kernel_ms_per_step'],r['frac'],len(d['per_op']))"
for w in googlenet-net nin-net alexnet-net; do
python bench.py --workload $w --dtype bf16 --layout nhwc --no-cpu-baseline --steps 20 --warmup 5 --graph --parallel-branches 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('direct=$ad $w',d['value'],d['images_per_s'],d['ms_per_step'],r['conv_ms'],r['non_conv_ms'],r['frac'])"
done; done
