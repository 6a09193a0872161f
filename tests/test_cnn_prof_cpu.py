"""BASELINE configs[0]: test/sgemm-ops-tiny.txt through the cnn-prof plumbing on the CPU (parse -> annotate -> signatures),
plus the CPU oracle executing the small op and checking itself."""
import os
import numpy as np
from boda_amd.cnn_prof import cnn_prof, main
from boda_amd.cnn_op import OpTune
from boda_amd.op import parse_op, read_ops
from oracle import boda_oracle as bo


def test_sgemm_ops_tiny_plumbing(golden_dir, tmp_path):
    fn = os.path.join(golden_dir, "ops", "sgemm-ops-tiny.txt")
    ops = read_ops(fn)                                   # legacy (type=...,dims_vals=...) text form
    assert [o.sgemm_geom()["M"] for o in ops] == [128, 2048, 8192]
    sigs = cnn_prof(ops, OpTune())
    assert all("func_name=hip_sgemm" in s for s in sigs)
    assert all("func_name=cublas_sgemm" in s for s in cnn_prof(ops, OpTune(use_culibs=1)))
    assert parse_op(sigs[0]).get_func_name() == "hip_sgemm"      # signatures are themselves valid op lines
    out = tmp_path / "sigs.txt"
    assert main(["--cnn-func-sigs-fn", fn, "--rtc-func-sigs-fn", str(out), "--op-tune", "(use_be=hip)"]) == 0
    assert out.read_text().splitlines() == sigs
    # reference CUCL variants are reported, not silently mapped
    assert cnn_prof(ops[:1], OpTune(use_be="ocl"))[0].startswith("# unsupported")
    # the CPU oracle executes the 128^3 op: exact-answer pattern (mode 600) and fp64-accumulate cross-check (mode 5)
    r = bo.run_op(ops[0], mode=600)
    m, n = np.meshgrid(np.arange(128), np.arange(128), indexing="ij")
    assert np.array_equal(r["c"], (1000 * m + n).astype(np.float32))
    r = bo.run_op(ops[0], mode=5)
    assert bo.mrd(bo.sgemm(r["a"], r["b"], f64acc=True), r["c"]) < 2e-4  # fp32 chain vs fp64 accumulate: inside the reference tolerance


def test_conv_signatures_carry_relu_and_native_name(golden_dir):
    ops = read_ops(os.path.join(golden_dir, "ops", "conv-ops-tiny.txt"))
    for s in cnn_prof(ops, OpTune()):
        o = parse_op(s)
        assert o.get_func_name() == "hip_conv" and o.get_u32("conv_has_relu") == 1
