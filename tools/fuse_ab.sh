for f in "" "--no-fuse-siblings"; do
  for g in "--graph --parallel-branches" "--graph" ""; do
    python bench.py --workload googlenet-net --dtype bf16 --layout nhwc --no-cpu-baseline --steps 20 --warmup 5 $g $f 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$f','$g',d['images_per_s'],d['ms_per_step'],d['roofline']['conv_ms'],d['roofline']['non_conv_ms'],len(d['per_call']))"
  done
done
for gs in "" "--group-siblings"; do
  python bench.py --workload googlenet --dtype bf16 --layout nhwc --no-cpu-baseline --steps 20 --warmup 5 --graph $gs 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('list $gs',d['value'],d['ms_per_step'],r['kernel_ms_per_step'],r['frac'],len(d['per_op']), {k:(v['ops'],v['ms'],v['frac']) for k,v in r['per_bound'].items()})"
done
