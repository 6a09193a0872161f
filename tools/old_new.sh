for d in _old . _old .; do
  (cd $d; python bench.py --workload googlenet --dtype bf16 --layout nhwc --no-cpu-baseline --steps 20 --warmup 5 --graph 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$d googlenet',d['value'],d['ms_per_step'],r['kernel_ms_per_step'],r['frac'])"
  python bench.py --workload resnet50 --dtype bf16 --layout nhwc --no-cpu-baseline --steps 20 --warmup 5 --graph 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$d resnet50',d['value'],d['ms_per_step'],r['kernel_ms_per_step'],r['frac'])")
done
