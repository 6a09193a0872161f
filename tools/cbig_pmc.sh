#!/bin/bash
# PMC probe of one layer under the staging-wave kernel (and the planner's kernel beside it): rocprofv3 --pmc on tools/cbig_probe.py, one counter set per run.
#   SPEC=alexnet:256 OPS=2 TILES=128x512x16x2x4x1x1x32x2x2 bash tools/cbig_pmc.sh
R=$PWD; cd /tmp && export TMPDIR=/tmp
i=0
for set in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_ACTIVE_INST_SCA"; do
  i=$((i+1)); O=$R/gpurun_out/cbig/pmc$i; rm -rf $O; mkdir -p $O
  rocprofv3 --kernel-trace --pmc $set -d $O -o p -- python $R/tools/cbig_probe.py --time ${SPEC:-alexnet:256} --ops ${OPS:-2} --tiles ${TILES:-128x512x16x2x4x1x1x32x2x2} --iters 3 > $O/log 2>&1
  python - <<PY
import sqlite3, glob
dbs = glob.glob("$O/*.db") + glob.glob("$O/*/*.db")
if not dbs: print("set $i: no db", open("$O/log").read()[-400:]); raise SystemExit
c = sqlite3.connect(dbs[0])
rows = c.execute("select kernel_name, grid_size, counter_name, sum(value), count(*) from counters_collection where kernel_name like 'bodahip_conv%' group by kernel_name, grid_size, counter_name").fetchall()
for kn, g, cn, v, n in rows: print(f"{kn[:28]:28s} grid {g:8d} {cn:30s} {v/n:16.1f}  (x{n})")
PY
  rm -rf $O
done
