python -m pytest tests/test_gpu_parity.py -x -q -k "row_decimated or conv_gen5 or golden or full_gen5" 2>&1 | tail -3
for r in "" 96x256x1x4x2 64x256x1x4x2 off; do echo "RDEC=$r"; BODAHIP_RDEC=$r python tools/tile_sweep.py --workload alexnet --ops 0 --iters 30 2>&1 | grep auto; done
for w in alexnet nin; do python bench.py --workload $w --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], [round(o['ms']*1e3,1) for o in d['per_op']][:3])"; done
