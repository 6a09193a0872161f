# GPU tool: fp32 nets with the pool taps as conditional / unconditional loads and LRN through powf / exp2-log2 (same box)
python -m pytest tests/test_gpu_fullnet.py -x -q -k "matches_oracle or bench_batch" 2>&1 | tail -3
for sp in 0 1; do
for w in googlenet-net nin-net alexnet-net; do
BODAHIP_SPEC_FWD=$sp python bench.py --workload $w --no-cpu-baseline --graph --parallel-branches 2>/dev/null | python -c "
import sys,json,collections; d=json.loads(sys.stdin.read()); r=d['roofline']; print('spec=$sp $w',d['value'],d['images_per_s'],d['ms_per_step'],r['conv_ms'],r['non_conv_ms'],r['frac'])
agg=collections.defaultdict(lambda:[0,0.0])
for c in d['per_call']:
    if not c['func'].startswith('hip_conv'): agg[c['func']][0]+=1; agg[c['func']][1]+=c['ms']
print('   ',{k:(v[0],round(v[1],4)) for k,v in agg.items()})"
done; done
