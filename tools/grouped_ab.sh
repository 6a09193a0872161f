for w in resnet50 googlenet; do for gs in "" "--group-siblings"; do
  python bench.py --workload $w --dtype bf16 --layout nhwc --no-cpu-baseline --steps 20 --warmup 5 --graph $gs 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$w $gs',d['value'],d['ms_per_step'],r['kernel_ms_per_step'],r['frac'],len(d['per_op']),r['timed_region'], d['config'].get('grouped_sibling_convs'))"
done; done
