"""-m gpu: N devices behind ONE rtc_compute_t (boda_amd/csrc/hip_multi.cc): vars with a leading `img` dim (sgemm: dim `M`) are
sharded, weights replicated, copy_nda_to_var scatters, run() enqueues on every device, copy_var_to_nda gathers.  The device list repeats GPU 0
({0,0}, {0,0,0}), so the sharding logic runs with the HIP kernels doing the arithmetic on a one-GPU box: gathered results must equal the
oracle -- and the single-device backend -- bit for bit, for even, uneven and empty shards."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from boda_amd.cnn_op import NATIVE_ARGS, OpTune, add_codegen_annotations
from boda_amd.op import Dims, Op, RtErr, UnsupErr, parse_op
from boda_amd.rtc import RtcArg, RtcFuncCall, RtcFuncInfo, make_rtc
from oracle import boda_oracle as bo


def _run(rtc, op, ins, tune=None):
    anno = add_codegen_annotations(op, tune or OpTune()); fn = anno.get_func_name()
    rtc.compile([RtcFuncInfo("f", "", [a for a, _ in NATIVE_ARGS[fn]], anno)])
    am, made = {}, []
    try:
        for an, io in NATIVE_ARGS[fn]:
            if io == "REF":
                am[an] = RtcArg.ref(anno.get_dims(an)); continue
            rtc.create_var_with_dims(an, anno.get_dims(an)); made.append(an); am[an] = RtcArg.var(an)
            assert rtc.get_var_dims(an) == anno.get_dims(an)       # the caller sees the logical dims
            if io == "IN":
                rtc.copy_nda_to_var(an, ins[an])
        ids = [rtc.run(RtcFuncCall("f", am)) for _ in range(2)]
        rtc.finish_and_sync()
        assert rtc.get_dur(ids[0], ids[1]) > 0
        outs = {an: rtc.copy_var_to_nda(an) for an, io in NATIVE_ARGS[fn] if io != "REF"}
        return outs
    finally:
        for vn in made:
            rtc.release_var(vn)
        rtc.release_func("f"); rtc.release_per_call_id_data()


def _conv_op(B, C, H, W, OC, KH, KW, S, P):
    OH = (H + 2 * P - KH) // S + 1; OW = (W + 2 * P - KW) // S + 1
    return parse_op(f"(str_vals=(type=Convolution),nda_vals=(biases=(dims=(out_chan={OC})),filts=(dims=(out_chan={OC},in_chan={C},y={KH},x={KW})),"
                    f"in=(dims=(img={B},chan={C},y={H},x={W})),in_pad=(tn=none,dims=(y={P},x={P})),kern_sz=(tn=none,dims=(y={KH},x={KW})),"
                    f"out=(dims=(img={B},chan={OC},y={OH},x={OW})),out_chans=(tn=uint32_t,v={OC}),stride=(tn=none,dims=(y={S},x={S}))))")


@pytest.fixture(scope="module", params=[2, 3])
def multi(request):
    r = make_rtc("(be=hip,devices=" + ":".join(["0"] * request.param) + ")")
    r.init()
    assert r.get_plat_tag().endswith(f"*{request.param}")
    yield r, request.param
    r.finish_and_sync(); r.close()


@pytest.fixture(scope="module")
def single():
    r = make_rtc("(be=hip)", 0); r.init()
    yield r
    r.finish_and_sync(); r.close()


@pytest.mark.parametrize("shape", [(5, 32, 14, 14, 64, 5, 5, 1, 2), (1, 19, 11, 11, 40, 1, 1, 1, 0), (8, 3, 35, 35, 96, 11, 11, 4, 0), (7, 64, 9, 9, 130, 3, 3, 1, 1), (2, 256, 6, 6, 512, 6, 6, 1, 0)])
def test_conv_sharded_over_devices_equals_unsharded(multi, single, shape):
    rtc, n = multi
    B, C, H, W, OC, KH, KW, S, P = shape
    ins = {"in": bo.gen_conv_in(B, C, H, W), "filts": bo.gen_conv_filts(OC, C, KH, KW), "biases": bo.gen_conv_biases(OC)}
    got = _run(rtc, _conv_op(*shape), ins)
    want = bo.conv_fwd(ins["in"], ins["filts"], ins["biases"], (S, S), (P, P), True)
    assert np.array_equal(got["out"], want)                                     # gathered shards == the oracle's unsharded result
    assert np.array_equal(got["out"], _run(single, _conv_op(*shape), ins)["out"])   # == the single-device backend
    for an in ("in", "filts", "biases"):                                        # scatter / broadcast followed by gather gives the inputs back
        assert np.array_equal(got[an], ins[an]), an


@pytest.mark.parametrize("M,N,K", [(100, 36, 50), (1, 64, 64), (257, 130, 70), (512, 512, 96)])
def test_sgemm_sharded_on_M(multi, single, M, N, K):
    """a is K:M (sharded along its SECOND dim: packed per device), b K:N replicated, c M:N sharded along its leading dim."""
    rtc, n = multi
    op = parse_op(f"(str_vals=(type=sgemm),nda_vals=(a=(dims=(K={K},M={M})),b=(dims=(K={K},N={N})),c=(dims=(M={M},N={N}))))")
    ins = {"a": bo.gen_sgemm_a(K, M), "b": bo.gen_sgemm_b(K, N)}
    got = _run(rtc, op, ins)
    assert np.array_equal(got["c"], bo.sgemm(ins["a"], ins["b"]))
    assert np.array_equal(got["a"], ins["a"]) and np.array_equal(got["b"], ins["b"])


def test_multi_device_contract(multi):
    rtc, n = multi
    # generated CUCL source: runs (on every device) on replicated vars, refused on sharded ones
    src = "CUCL_GLOBAL_KERNEL void add1( GASQ float * const a, uint32_t const n ) { if( GLOB_ID_1D < n ) { a[GLOB_ID_1D] += 1.0f; } }\n"
    rtc.compile([RtcFuncInfo("add1", src, ["a", "n"], Op({"type": "x", "func_name": "add1"}, {}))])
    rtc.create_var_with_dims("r", Dims(("v",), (1000,), "float"))
    rtc.create_var_with_dims("s", Dims(("img", "chan"), (10, 100), "float"))
    try:
        rtc.copy_nda_to_var("r", np.arange(1000, dtype=np.float32))
        rtc.run(RtcFuncCall("add1", {"a": RtcArg.var("r"), "n": RtcArg.scalar(1000, "uint32_t")}, tpb=256, blks=4))
        rtc.finish_and_sync()
        assert np.array_equal(rtc.copy_var_to_nda("r"), np.arange(1000, dtype=np.float32) + 1)
        with pytest.raises(UnsupErr):
            rtc.run(RtcFuncCall("add1", {"a": RtcArg.var("s"), "n": RtcArg.scalar(1000, "uint32_t")}, tpb=256, blks=4))
        with pytest.raises(RtErr):
            rtc.get_var_raw_native_pointer("s")            # a sharded var has no single device pointer
        assert rtc.get_var_raw_native_pointer("r") != 0
        x = np.arange(1000, dtype=np.float32).reshape(10, 100)
        rtc.copy_nda_to_var("s", x)
        assert np.array_equal(rtc.copy_var_to_nda("s"), x)
        rtc.set_var_to_zero("s")
        assert not rtc.copy_var_to_nda("s").any()
    finally:
        rtc.release_var("r"); rtc.release_var("s"); rtc.release_func("add1"); rtc.release_per_call_id_data()
