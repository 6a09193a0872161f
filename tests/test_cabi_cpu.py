"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/bodahip.h declares,
reports a missing GPU loudly (no fallback), and its hiprtc path cross-compiles the kernel templates for gfx950."""
import os
import re
import ctypes
import pytest

from boda_amd import rtc as R
from boda_amd.op import RtErr, UnsupErr, Dims
from boda_amd import gen_data as gd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "bodahip.h")).read()
    declared = set(re.findall(r"\b(bodahip_[a-z_0-9]+)\s*\(", hdr))
    assert len(declared) >= 29
    lib = ctypes.CDLL(R.SO_PATH)
    for sym in sorted(declared):
        assert hasattr(lib, sym), f"libbodahip.so does not export {sym}"
    assert declared == set(R.ABI), "python binding table out of sync with the header"
    assert lib.bodahip_abi_version() == 1


def test_no_gpu_is_a_loud_error_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = R.HipCompute(0)
    with pytest.raises(RtErr, match="no HIP device"):
        h.init()
    h.close()


def test_unknown_backend_rejected():
    with pytest.raises(RtErr):
        R.make_rtc("(be=nvrtc)")


def test_cucl_prelude_compiles_generic_source_offline():
    assert R.compile_offline(gd.SRC) > 1000
    with pytest.raises(RtErr, match="HIPRTC_ERROR_COMPILATION"):
        R.compile_offline("CUCL_GLOBAL_KERNEL void f( GASQ float * a ) { a[GLOB_ID_1D] = undefined_symbol; }")


@pytest.mark.parametrize("opts", [
    "-DBI=128 -DBJ=128 -DBK=16 -DWI=2 -DWJ=2 -DMINW=2 -DI_MODE=0 -DJ_MODE=0 -DEPI=0",
    "-DBI=64 -DBJ=64 -DBK=16 -DWI=1 -DWJ=1 -DMINW=1 -DI_MODE=1 -DJ_MODE=1 -DEPI=0",
    "-DBI=96 -DBJ=128 -DBK=16 -DWI=1 -DWJ=2 -DMINW=2 -DI_MODE=3 -DJ_MODE=2 -DEPI=1 -DKH=11 -DKW=11 -DSY=4 -DSX=4 -DPY=0 -DPX=0 -DRELU=1",
    "-DBI=128 -DBJ=128 -DBK=16 -DWI=2 -DWJ=2 -DMINW=2 -DI_MODE=2 -DJ_MODE=2 -DEPI=1 -DKH=3 -DKW=3 -DSY=1 -DSX=1 -DPY=1 -DPX=1 -DRELU=1",
])
def test_native_kernel_template_cross_compiles_for_gfx950(opts):
    assert R.compile_offline(opts + " -DKNAME=k_test", "gemm_conv_f32") > 4000


def test_cpp_op_parser_agrees_with_python_on_every_fixture_line(golden_dir):
    """csrc/lexp.cc (what bodahip_compile uses on rtc_func_info_t.op) == boda_amd/op.py on all fixture op lines, both text forms."""
    import glob
    from boda_amd.op import parse_op
    n = 0
    for fn in sorted(glob.glob(os.path.join(golden_dir, "ops", "*.txt"))):
        for line in open(fn):
            if line.strip():
                assert R.parse_op_native(line) == parse_op(line).to_str(), (fn, line[:80])
                n += 1
    assert n > 400
    with pytest.raises(RtErr):
        R.parse_op_native("(str_vals=(type=sgemm),nda_vals=(a=(dims=(K=2,M=3))")
    with pytest.raises(RtErr):
        R.parse_op_native("(bogus=1)")
    # annotated ops (func_name, uint32 scalars with values) round-trip too
    s = "(str_vals=(func_name=hip_conv,type=Convolution),nda_vals=(conv_has_relu=(tn=uint32_t,v=1),stride=(tn=none,dims=(y=4,x=4))))"
    assert R.parse_op_native(s) == s
