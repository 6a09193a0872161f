"""GPU tool: the staging-wave sgemm kernel (kernels/sgemm_big_f32.hip) under its tile forms against the planner's choice, sizes of test/sgemm-ops-full.txt.
   python tools/sgemm_big_sweep.py [sizes ...]      us / TF/s per (size, tile); up to 3072^3 the result is compared bit for bit with the planner's kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from boda_amd.cnn_op import OpTune, add_codegen_annotations
from boda_amd.ops_prof import OpsBackend, profile_rcg_call
from boda_amd.rtc import make_rtc
sizes = [int(x) for x in sys.argv[1:]] or [1024, 1536, 2048, 3072, 4096, 5120, 6144]
TILES = os.environ.get("TILES", "auto 128x128x8x3x4x2 128x128x16x3x4x2 128x128x8x3x4x1 256x128x8x3x4x1 128x256x8x3x4x1 256x128x16x3x4x1 128x256x16x3x4x1 256x256x8x3x4x1").split()
by_n = {op.sgemm_geom()["M"]: op for op in bench.sgemm_full_ops()}
rtc = make_rtc("(be=hip)", 0); rtc.init(); be = OpsBackend(rtc)
nbad = 0
for n in sizes:
    op = by_n[n]; base = None
    for t in TILES:
        tt = "" if t == "auto" else t
        try:
            anno = add_codegen_annotations(op, OpTune(hip_tile=tt))
            check = n <= 3072
            outs, prc = profile_rcg_call(be, anno, 5, 0.0, 8, want_outs=check, tile=tt)
            us = float(np.median(prc.all_secs[2:])) * 1e6
            same = ""
            if check:
                if base is None: base = outs["c"]
                else:
                    eq = bool(np.array_equal(base, outs["c"])); same = "  bit-identical" if eq else "  <-- DIFFERS"; nbad += 0 if eq else 1
            print(f"{n:6d}^3 {t:22s} [{prc.launch['kernel'][8:]:14s} {prc.launch['cfg']:22s} grid {prc.launch['grid']:5d}] {us:9.1f} us {2.0*n**3/us/1e6:7.1f} TF/s{same}", flush=True)
        except Exception as e:
            print(f"{n:6d}^3 {t:22s} ERR {type(e).__name__}: {str(e)[:120]}", flush=True)
print("differing results:", nbad)
rtc.close()
