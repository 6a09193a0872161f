#!/bin/bash
# clock stamps of conv_big_f32.hip launches IN THE LAYER SEQUENCE (bench.py --workload W with a tile-wisdom table):  cbig_tl_seq.sh <workload> <batch> "<op>=<tile>[;...]"
cd "$(dirname "$0")/.."
W=$1; B=$2; SPEC=$3; F=/tmp/wis_tl_$$.txt; TS=/tmp/cbig_ts_$$.txt; rm -f $TS
python - "$W" "$B" "$SPEC" > $F <<'P'
import sys, bench
w, b, spec = sys.argv[1], int(sys.argv[2]), sys.argv[3]
ops = bench.alexnet_b256_ops(b) if w == "alexnet" else bench.nin_ops(b)
for kv in spec.split(";"):
    i, t = kv.split("="); print(f"{ops[int(i)].to_str()}\t{t}\t0\t0")
P
N=$(echo "$SPEC" | tr ';' '\n' | wc -l)
BODAHIP_TILE_WISDOM=$F BODAHIP_CBIG_TSTAMP=$TS${LATE:+:late} BODAHIP_EXTRA_DEFS="-DTSTAMP=1" python bench.py --workload $W --batch $B --steps ${STEPS:-4} --warmup 2 --no-cpu-baseline > /dev/null 2>${TLERR:-/dev/null}
python tools/cbig_tl_parse.py $TS $N
