for t in call kernel; do for w in googlenet resnet50; do
  python bench.py --workload $w --dtype bf16 --layout nhwc --no-cpu-baseline --steps 20 --warmup 5 --graph --timing $t 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$w $t',d['value'],d['ms_per_step'],r['kernel_ms_per_step'],r['frac'],r['timed_region']['frac'], [round(p['ms']*1e3,1) for p in d['per_op']][:12])"
done; done
python bench.py --workload alexnet --no-cpu-baseline --timing call 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('alexnet call', d['value'], d['roofline']['frac'], d['roofline']['kernel_ms_per_step'])"
python bench.py --workload alexnet --no-cpu-baseline --timing kernel 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('alexnet kernel', d['value'], d['roofline']['frac'], d['roofline']['kernel_ms_per_step'])"
python bench.py --workload alexnet --exact 0 --no-cpu-baseline --timing kernel 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('alexnet tol kernel', d['value'], d['roofline']['frac'], d['roofline']['kernel_ms_per_step'])"
