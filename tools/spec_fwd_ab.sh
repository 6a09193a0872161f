# GPU tool: channels-last nets with the generic / geometry-specialised pool + LRN kernels (same box)
python -m pytest tests/test_gpu_fullnet.py -x -q -k "specialised or channels_last" 2>&1 | tail -3
for sp in 0 1; do
for w in googlenet-net nin-net alexnet-net; do
BODAHIP_SPEC_FWD=$sp python bench.py --workload $w --dtype bf16 --layout nhwc --no-cpu-baseline --steps 20 --warmup 5 --graph --parallel-branches 2>/dev/null | python -c "
import sys,json,collections; d=json.loads(sys.stdin.read()); r=d['roofline']; print('spec=$sp $w',d['value'],d['images_per_s'],d['ms_per_step'],r['conv_ms'],r['non_conv_ms'],r['frac'])
agg=collections.defaultdict(lambda:[0,0.0])
for c in d['per_call']:
    if not c['func'].startswith('hip_conv'): agg[c['func']][0]+=1; agg[c['func']][1]+=c['ms']
print('   ',{k:(v[0],round(v[1],4)) for k,v in agg.items()})"
done; done
