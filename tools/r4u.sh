for r in off ""; do for w in alexnet nin; do echo "RDEC=$r $w"; BODAHIP_RDEC=$r python bench.py --workload $w --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], [round(o['ms']*1e3,1) for o in d['per_op']][:3])"; done; done
