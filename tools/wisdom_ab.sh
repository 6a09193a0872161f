#!/bin/bash
# GPU tool: in-sequence A/B of per-op tiles through the tile-wisdom path (BODAHIP_TILE_WISDOM): bench.py --workload <w> with / without the table, alternating.
#   tools/wisdom_ab.sh <workload> <batch> "<op index>=<tile>[;<op index>=<tile>...]"
cd "$(dirname "$0")/.."
W=$1; B=$2; SPEC=$3; F=/tmp/wis_ab_$$.txt
python - "$W" "$B" "$SPEC" > $F <<'P'
import sys, bench
w, b, spec = sys.argv[1], int(sys.argv[2]), sys.argv[3]
ops = bench.alexnet_b256_ops(b) if w == "alexnet" else bench.nin_ops(b)
for kv in spec.split(";"):
    i, t = kv.split("="); print(f"{ops[int(i)].to_str()}\t{t}\t0\t0")
P
run() { env $2 python bench.py --workload $W --batch $B --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('%-8s' % '$1', 'ms/step %.4f  frac %.4f |' % (d['ms_per_step'], d['roofline']['frac']), ' '.join('%.0f' % (o['ms'] * 1e3) for o in d['per_op']))
"; }
for i in 1 2 3; do run base "X=1"; run wisdom "BODAHIP_TILE_WISDOM=$F"; done
