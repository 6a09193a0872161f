cd /root/repo; mkdir -p gpurun_out/knobs
run() { env $2 python bench.py --workload alexnet --batch 256 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('%-34s' % '$1', 'ms/step %.4f  frac %.4f |' % (d['ms_per_step'], d['roofline']['frac']), ' '.join('%.0f' % (o['ms'] * 1e3) for o in d['per_op']), '| cache', d.get('compile', {}).get('cache', ''))
"; }
( for i in 1 2; do run default X=1; for n in 1 2 3 4; do run NOOP=$n BODAHIP_EXTRA_DEFS=-DNOOP=$n; done; done
  echo "== fresh cache dir for the default plan (compiled on this box, then served from that cache)"
  for i in 1 2 3; do run default-freshcache BODAHIP_CACHE_DIR=/tmp/kc_fresh; done
  for i in 1 2; do run default X=1; done ) 2>&1 | tee gpurun_out/knobs/log3.txt
