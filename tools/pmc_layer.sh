#!/bin/bash
# one-off PMC probe of ONE layer: rocprofv3 --pmc on tools/nhwc_sweep.py (NET, SEL, BATCH, TILES from the environment); one counter set per run
R=$PWD; cd /tmp && export TMPDIR=/tmp
i=0
for set in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_ACTIVE_INST_SCA" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_BUSY_avr TA_TA_BUSY_sum"; do
  i=$((i+1)); O=$R/gpurun_out/pmc_layer/$i; rm -rf $O; mkdir -p $O
  rocprofv3 --kernel-trace --pmc $set -d $O -o p -- python $R/tools/nhwc_sweep.py ${NET:-alexnet} 4 > $O/log 2>&1
  python - <<PY
import sqlite3, glob
dbs = glob.glob("$O/*.db")
if not dbs: print("set $i: no db", open("$O/log").read()[-400:]); raise SystemExit
c = sqlite3.connect(dbs[0])
rows = c.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection where kernel_name like 'bodahip_conv%' group by kernel_name, counter_name").fetchall()
for kn, cn, v, n in rows: print(f"{kn[:34]:34s} {cn:36s} {v/n:16.1f}  (x{n})")
PY
done
