"""-m gpu: the cnn_op_info mode end to end (src/cnn-prof.cc:24-130): ops annotated, profiled on be=hip with inputs generated on the device, profiled
again on a comparison backend under op_tune_comp (the reference's default: use_culibs=1 -- here the cudnn_conv / cublas_sgemm aliases of a second
be=hip instance) and compared var by var, info + efficiency rows written in the reference's row text."""
import io
import os
import re

import pytest

pytestmark = pytest.mark.gpu

from boda_amd.cnn_op import OpTune
from boda_amd.cnn_op_info import cnn_op_info
from boda_amd.op import read_ops
from boda_amd.rtc import make_rtc


def test_cnn_op_info_rows_and_comparison(golden_dir):
    ops = read_ops(os.path.join(golden_dir, "ops", "conv-ops-debug.txt")) + read_ops(os.path.join(golden_dir, "ops", "sgemm-ops-tiny.txt"))[:2]
    rtc = make_rtc("(be=hip)", 0); rtc.init()
    cpu = make_rtc("(be=hip)", 0); cpu.init()      # the comparison backend (--rtc-comp)
    try:
        out, info, eff = io.StringIO(), io.StringIO(), io.StringIO()
        n = cnn_op_info(rtc, ops, OpTune(), gen_mode=5, rtc_comp=cpu, op_tune_comp=OpTune(use_culibs=1), out=out, info_out=info, eff_out=eff, inc_op_info_in_eff=1)
        assert n == 0 and out.getvalue().rstrip().endswith("***ALL IS WELL***") and out.getvalue().count("vars_to_compare: ['out']") == len(ops) - 2
        rows = eff.getvalue().splitlines()
        assert len(rows) == len(ops) and all(r.endswith("\\\\ ") for r in rows)
        conv = [r for r in rows if "\\verb|Convolution|" in r]
        assert len(conv) == len(ops) - 2
        for r in conv:      # KSZ & Stride & out_chans & $dims(in)$ & variant & MxKxN & Bytes & FLOPs & F/B & Runtime & F/s & %Peak
            f = [x.strip() for x in r[:-3].split("&")]
            assert len(f) == 12 and re.fullmatch(r"[0-9.]+[mun]s", f[9]) and re.fullmatch(r"[0-9.]+[GT]F/s", f[10]) and 0 < float(f[11].rstrip("m")) 
        sg = [r for r in rows if "\\verb|" not in r]
        assert len(sg) == 2 and all(r.rstrip("\\ ").endswith("x") for r in sg)      # ... & speedup-of-non-comp
        assert len(info.getvalue().splitlines()) == len(ops)
        # a comparison that must fail is counted: tolerance 0 against a backend with another summation order is not needed -- mismatching data is
        bad = io.StringIO()
        n_bad = cnn_op_info(rtc, ops[:1], OpTune(hip_dtype="bf16"), gen_mode=5, rtc_comp=cpu, op_tune_comp=OpTune(use_culibs=1), mrd_toler=1e-7, out=bad)
        assert n_bad == 1 and "***MAD FAILS*** num_mad_fail=1" in bad.getvalue()
    finally:
        rtc.close(); cpu.close()
