#!/bin/bash
# GPU tool: clock stamps of every staging-wave launch of one pass over a layer list (per-launch synchronisation: phases inside a workgroup, not the sequence's clocks), for
# the layer whose tile matches <grep>.   tools/cbig_tl_conv1.sh <workload> <batch> "<op>=<tile>" <grep>
cd "$(dirname "$0")/.."
W=$1; B=$2; SPEC=$3; G=$4; F=/tmp/wis_tlc_$$.txt; TS=/tmp/cbig_tsc_$$.txt; rm -f $TS
python - "$W" "$B" "$SPEC" > $F <<'P'
import sys, bench
w, b, spec = sys.argv[1], int(sys.argv[2]), sys.argv[3]
ops = bench.alexnet_b256_ops(b) if w == "alexnet" else bench.nin_ops(b)
for kv in spec.split(";"):
    i, t = kv.split("="); print(f"{ops[int(i)].to_str()}\t{t}\t0\t0")
P
BODAHIP_TILE_WISDOM=$F BODAHIP_CBIG_TSTAMP=$TS BODAHIP_EXTRA_DEFS="-DTSTAMP=1" timeout 200 python bench.py --workload $W --batch $B --steps 2 --warmup 1 --settle-ms 0 --no-cpu-baseline > /dev/null 2>&1
python tools/cbig_tl_parse.py $TS 200 | grep "$G" | tail -2
