for w in resnet50 googlenet; do for f in "" "--no-patch"; do
  python bench.py --workload $w --dtype bf16 --layout nhwc --no-cpu-baseline --steps 20 --warmup 5 --graph $f 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$w $f',d['value'],d['ms_per_step'],r['kernel_ms_per_step'],r['frac'],{k:(v['ops'],v['ms'],v['frac']) for k,v in r['per_bound'].items()}); print('   ',[round(p['ms']*1e3,1) for p in d['per_op']])"
done; done
python bench.py --workload googlenet --dtype bf16 --layout nhwc --no-cpu-baseline --steps 20 --warmup 5 --graph --group-siblings 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('googlenet grouped',d['value'],d['ms_per_step'],r['kernel_ms_per_step'],r['frac'],len(d['per_op']))"
for w in googlenet-net nin-net alexnet-net; do for f in "" "--no-patch"; do
python bench.py --workload $w --dtype bf16 --layout nhwc --no-cpu-baseline --steps 20 --warmup 5 --graph --parallel-branches $f 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$w $f',d['value'],d['images_per_s'],d['ms_per_step'],r['conv_ms'],r['non_conv_ms'],r['frac'])"
done; done
