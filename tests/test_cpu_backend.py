"""be=cpu (boda_amd/csrc/cpu_compute.cc): the host-cores backend behind the same rtc_compute_t contract and C ABI as be=hip -- the CPU baseline
SURVEY.md section 8(d) asks for (blocked, vectorised, OpenMP sgemm / conv + bias + ReLU on reference-layout tensors).  Its outputs are one
ascending-k fp32 fma chain per element, like the reference's kernels: BIT-EXACT against the oracle (and hence equal to be=hip's fp32 path)."""
import numpy as np
import pytest

from boda_amd.cnn_op import NATIVE_ARGS, OpTune, add_codegen_annotations
from boda_amd.op import Dims, Op, RtErr, UnsupErr, parse_op
from boda_amd.rtc import RtcArg, RtcFuncCall, RtcFuncInfo, make_rtc
from oracle import boda_oracle as bo


@pytest.fixture(scope="module", params=["avx512", "avx2"])
def rtc(request, monkeypatch_module=None):
    import os
    old = os.environ.pop("BODACPU_NO_AVX512", None)
    if request.param == "avx2":
        os.environ["BODACPU_NO_AVX512"] = "1"
    r = make_rtc("(be=cpu)")
    try:
        r.init()
    finally:
        os.environ.pop("BODACPU_NO_AVX512", None)
        if old is not None:
            os.environ["BODACPU_NO_AVX512"] = old
    tag = r.get_plat_tag()
    assert tag.startswith("cpu:")
    if request.param == "avx512" and ":avx512:" not in tag:
        pytest.skip("host has no AVX-512")
    yield r
    r.close()


def _run(rtc, op, ins, tune=None):
    anno = add_codegen_annotations(op, tune or OpTune()); fn = anno.get_func_name()
    rtc.compile([RtcFuncInfo("f", "", [a for a, _ in NATIVE_ARGS[fn]], anno)])
    am, made = {}, []
    try:
        for an, io in NATIVE_ARGS[fn]:
            if io == "REF":
                am[an] = RtcArg.ref(anno.get_dims(an)); continue
            rtc.create_var_with_dims(an, anno.get_dims(an)); made.append(an); am[an] = RtcArg.var(an)
            if io == "IN":
                rtc.copy_nda_to_var(an, ins[an])
        cid = rtc.run(RtcFuncCall("f", am))
        rtc.finish_and_sync()
        assert rtc.get_dur(cid, cid) > 0
        return rtc.copy_var_to_nda([a for a, io in NATIVE_ARGS[fn] if io == "OUT"][0])
    finally:
        for vn in made:
            rtc.release_var(vn)
        rtc.release_func("f"); rtc.release_per_call_id_data()


def _sgemm_op(M, N, K):
    return parse_op(f"(str_vals=(type=sgemm),nda_vals=(a=(dims=(K={K},M={M})),b=(dims=(K={K},N={N})),c=(dims=(M={M},N={N}))))")


def _conv_op(B, C, H, W, OC, KH, KW, S, P):
    OH = (H + 2 * P - KH) // S + 1; OW = (W + 2 * P - KW) // S + 1
    return parse_op(f"(str_vals=(type=Convolution),nda_vals=(biases=(dims=(out_chan={OC})),filts=(dims=(out_chan={OC},in_chan={C},y={KH},x={KW})),"
                    f"in=(dims=(img={B},chan={C},y={H},x={W})),in_pad=(tn=none,dims=(y={P},x={P})),kern_sz=(tn=none,dims=(y={KH},x={KW})),"
                    f"out=(dims=(img={B},chan={OC},y={OH},x={OW})),out_chans=(tn=uint32_t,v={OC}),stride=(tn=none,dims=(y={S},x={S}))))")


@pytest.mark.parametrize("M,N,K", [(64, 64, 64), (100, 36, 50), (33, 257, 19), (1, 1, 1), (97, 513, 300), (384, 384, 384), (7, 1000, 1025)])
def test_cpu_sgemm_bit_exact(rtc, M, N, K):
    a = bo.gen_sgemm_a(K, M); b = bo.gen_sgemm_b(K, N)
    assert np.array_equal(_run(rtc, _sgemm_op(M, N, K), {"a": a, "b": b}), bo.sgemm(a, b))


@pytest.mark.parametrize("shape", [(2, 19, 11, 11, 40, 1, 1, 1, 0), (3, 64, 14, 14, 128, 1, 1, 2, 0), (2, 8, 7, 7, 16, 1, 1, 1, 1), (1, 3, 12, 12, 16, 3, 3, 1, 1),
                                   (2, 5, 17, 13, 7, 5, 5, 2, 2), (1, 8, 6, 6, 40, 6, 6, 1, 0), (2, 3, 35, 35, 96, 11, 11, 4, 0), (1, 16, 14, 14, 130, 7, 7, 2, 3),
                                   (1, 1, 5, 5, 1, 5, 5, 1, 2), (3, 4, 33, 31, 20, 5, 5, 1, 2), (9, 300, 1, 1, 100, 1, 1, 1, 0)])
def test_cpu_conv_bit_exact(rtc, shape):
    B, C, H, W, OC, KH, KW, S, P = shape
    i = bo.gen_conv_in(B, C, H, W); f = bo.gen_conv_filts(OC, C, KH, KW); bi = bo.gen_conv_biases(OC)
    got = _run(rtc, _conv_op(*shape), {"in": i, "filts": f, "biases": bi})
    assert np.array_equal(got, bo.conv_fwd(i, f, bi, (S, S), (P, P), True))
    # the reference's door names land on the same kernels
    got2 = _run(rtc, _conv_op(*shape), {"in": i, "filts": f, "biases": bi}, OpTune(use_culibs=1))
    assert np.array_equal(got2, got)


def test_cpu_backend_contract(rtc):
    """vars are zero-filled, views share storage, generated CUCL source is refused as unsupported (recordable), unknown names are fatal."""
    d = Dims(("v",), (1000,), "float")
    rtc.create_var_with_dims("z", d)
    assert not rtc.copy_var_to_nda("z").any()
    rtc.create_var_with_dims_as_reshaped_view_of_var("zv", Dims(("a", "b"), (10, 100), "float"), "z")
    rtc.copy_nda_to_var("zv", np.arange(1000, dtype=np.float32).reshape(10, 100))
    assert np.array_equal(rtc.copy_var_to_nda("z"), np.arange(1000, dtype=np.float32))
    with pytest.raises(RtErr):
        rtc.create_var_with_dims("z", d)
    rtc.release_var("zv"); rtc.release_var("z")
    with pytest.raises(UnsupErr):
        rtc.compile([RtcFuncInfo("k", "CUCL_GLOBAL_KERNEL void k( GASQ float * const a ) { a[GLOB_ID_1D] = 1.0f; }", ["a"], Op({"type": "x", "func_name": "k"}, {}))])
    with pytest.raises(RtErr):
        rtc.run(RtcFuncCall("nope", {}))
    with pytest.raises(RtErr):
        make_rtc("(be=opencl)")
