#!/bin/bash
# PMC probe of the AlexNet fc6 launch (tools/tile_sweep.py --ops 5) under BODAHIP_FC / BODAHIP_EXTRA_DEFS from the environment; one counter set per run
R=$PWD; cd /tmp && export TMPDIR=/tmp
i=0
for set in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_ACTIVE_INST_SCA" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1)); O=$R/gpurun_out/pmc_fc/$i; rm -rf $O; mkdir -p $O
  rocprofv3 --kernel-trace --pmc $set -d $O -o p -- python $R/tools/tile_sweep.py --workload alexnet --ops ${OPS:-5} --iters 3 > $O/log 2>&1
  python - <<PY
import sqlite3, glob
dbs = glob.glob("$O/*.db")
if not dbs: print("set $i: no db", open("$O/log").read()[-400:]); raise SystemExit
c = sqlite3.connect(dbs[0])
try:
    rows = c.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection where kernel_name like 'bodahip_%' group by kernel_name, counter_name").fetchall()
    for kn, cn, v, n in rows: print(f"{kn[:28]:28s} {cn:36s} {v/n:16.1f}  (x{n})")
except Exception as e: print("query failed", e, open("$O/log").read()[-300:])
PY
done
