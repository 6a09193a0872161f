#!/bin/bash
# GPU tool: shader clock (GRBM_GUI_ACTIVE / 8 XCDs / kernel duration) and MFMA-busy share of one layer under the patch kernel's ablations
R=$PWD; cd /tmp && export TMPDIR=/tmp
for ab in 0 1 2; do
  O=$R/gpurun_out/clock_probe/$ab; rm -rf $O; mkdir -p $O
  BODAHIP_EXTRA_DEFS="-DABLATE=$ab" BODAHIP_CACHE_DIR=/tmp/kc_cp$ab rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d $O -o p -- python $R/tools/nhwc_sweep.py ${NET:-alexnet} 6 > $O/log 2>&1
  python - <<PY
import sqlite3, glob
c = sqlite3.connect(glob.glob("$O/*.db")[0])
rows = c.execute("select counter_name, sum(value), count(*), sum(end-start) from counters_collection where kernel_name like 'bodahip_conv_nhwc_patch%' group by counter_name").fetchall()
d = {r[0]: (r[1] / r[2], r[3] / r[2]) for r in rows}
gui, dur = d["GRBM_GUI_ACTIVE"]
print(f"ABLATE=$ab: duration {dur/1e3:7.1f} us  cycles/XCD {gui/8:9.0f}  clock {gui/8/dur:5.2f} GHz  MFMA busy {100*d['SQ_VALU_MFMA_BUSY_CYCLES'][0]/1024/(gui/8):5.1f} % of cycles")
PY
done
