mkdir -p gpurun_out/s5
python -m pytest tests/test_gpu_parity.py -x -q -k "fc_kernel or k1_stream" 2>&1 | tail -3
python bench.py --workload alexnet --no-cpu-baseline > gpurun_out/s5/alexnet.json 2>gpurun_out/s5/alexnet.err; python bench.py --workload nin --no-cpu-baseline > gpurun_out/s5/nin.json 2>/dev/null
python - <<'PY'
import json
for w in ('alexnet','nin'):
    d=json.loads(open(f'gpurun_out/s5/{w}.json').read().strip().splitlines()[-1])
    print(w, d['value'], d['ms_per_step'], d['roofline']['frac'], [round(o['ms']*1e3,1) for o in d['per_op']])
PY
