cd /root/repo; mkdir -p gpurun_out/prio
(
for p in 1 2 3; do
for w in "alexnet 256" "nin 256" "nin 128"; do echo "== $w STGPRIO=$p (switched) vs default"; timeout 300 bash tools/env_ab_ops.sh "BODAHIP_EXTRA_DEFS=-DSTGPRIO=$p" $w 2; done
done
for i in 1 2; do for e in X=1 BODAHIP_EXTRA_DEFS=-DSTGPRIO=1 BODAHIP_EXTRA_DEFS=-DSTGPRIO=2 BODAHIP_EXTRA_DEFS=-DSTGPRIO=3; do env $e timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-conv-ops 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('%-32s %.2f TF/s %.3f ms |' % ('$e', d['value'], d['ms_per_step']), ' '.join('%.1f' % o['tflops'] for o in d['per_op'][5:]))
"; done; done
) 2>&1 | tee gpurun_out/prio/log.txt
