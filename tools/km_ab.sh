cd /root/repo; mkdir -p gpurun_out/km
( timeout 900 python -m pytest tests/test_gpu_fullnet.py -x -q -m gpu -k "k_major_once or k1_chains or forward_matches_oracle" 2>&1 | tail -5
for i in 1 2 3; do
for e in "X=1" "BODAHIP_FILTS_KM_ONCE=off"; do
 env $e python bench.py --workload nin-net --batch 128 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('$e', 'ms/step %.4f  frac %.4f value %.1f' % (d['ms_per_step'], d['roofline']['frac'], d['value']))"
done; done ) > gpurun_out/km/log.txt 2>&1
