"""Parity tests proper (-m gpu): the HIP path, called through the C ABI, against
  (1) the reference's own golden digests (tests/golden/wisdom/*.wis) -- every one of the 251,
  (2) the CPU oracle on the same deterministic inputs (bit-exact: both are fp32 fma chains in ascending k),
  (3) size-independent properties at BASELINE.json's full sizes (exact-answer sgemm, batch-prefix invariance).
Tolerance where a tolerance applies: max-rel-diff < 2e-4 (src/rtc_prof.cc:161), digest checksums scaled as
src/boda_base.cc:306-307.  Nothing here reads /root/reference."""
import os
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from boda_amd import gen_data as gd
from boda_amd.cnn_op import NATIVE_ARGS, OpTune, add_codegen_annotations
from boda_amd.digest import Digest, read_wisdoms, SsdsDiff
from boda_amd.op import Dims, Op, RtErr, UnsupErr, parse_op, read_ops
from boda_amd.ops_prof import OpsBackend, ops_prof, profile_rcg_call
from boda_amd.rtc import HipCompute, RtcArg, RtcFuncCall, RtcFuncInfo, make_rtc
from oracle import boda_oracle as bo

MRD = 2e-4


MRD_REASSOC = 2e-3  # the reference's bound for kernels that re-associate the sum (cuDNN Winograd, src/rtc_prof.cc:317-319)


def _assert_matches_oracle(want, got, launch):
    """Default kernels keep the reference's ascending-k fp32 fma chain: bit-exact (and hence < 2e-4).  The explicit
    split-K tune re-associates the sum; it is held to the reference's own bound for re-associating kernels, 2e-3."""
    sd = SsdsDiff.of(want, got)
    assert not sd.has_nan()
    if "_s" in launch["cfg"]:
        assert sd.mrd < MRD_REASSOC, sd.basic_str()
    else:
        assert sd.mrd < MRD, sd.basic_str()
        assert np.array_equal(want, got), (launch["cfg"], sd.basic_str())


@pytest.fixture(scope="module")
def be():
    rtc = make_rtc("(be=hip)", 0)
    rtc.init()
    assert rtc.get_plat_tag().startswith("hip:")
    b = OpsBackend(rtc)
    yield b
    rtc.finish_and_sync()
    rtc.close()


def _sgemm_op(M, N, K):
    return parse_op(f"(str_vals=(type=sgemm),nda_vals=(a=(dims=(K={K},M={M})),b=(dims=(K={K},N={N})),c=(dims=(M={M},N={N}))))")


def _conv_op(B, C, H, W, OC, KH, KW, S, P):
    OH = (H + 2 * P - KH) // S + 1; OW = (W + 2 * P - KW) // S + 1
    return parse_op(f"(str_vals=(type=Convolution),nda_vals=(biases=(dims=(out_chan={OC})),filts=(dims=(out_chan={OC},in_chan={C},y={KH},x={KW})),"
                    f"in=(dims=(img={B},chan={C},y={H},x={W})),in_pad=(tn=none,dims=(y={P},x={P})),kern_sz=(tn=none,dims=(y={KH},x={KW})),"
                    f"out=(dims=(img={B},chan={OC},y={OH},x={OW})),out_chans=(tn=uint32_t,v={OC}),stride=(tn=none,dims=(y={S},x={S}))))")


def _run(be, op, mode=5, tune=None, include_ins=False, run_iter=1):
    t = tune or OpTune()
    anno = add_codegen_annotations(op, t)
    outs, prc = profile_rcg_call(be, anno, mode, 0.0, run_iter, include_ins=include_ins, tile=t.hip_tile)
    return outs, prc


# ---------------------------------------------------------------------------------------------------------------
# backend contract (the reference's rtc_test mode, src/rtc_compute.cc:135-194, plus var semantics)
# ---------------------------------------------------------------------------------------------------------------
DOT_SRC = """
CUCL_GLOBAL_KERNEL void my_dot( GASQ float const * const a, GASQ float const * const b, GASQ float * const c, uint32_t const n ) {
  uint32_t const ix = GLOB_ID_1D;
  if( ix < n ) { c[ix] = a[ix] + b[ix]; }
}
struct n_t { uint32_t n; };
CUCL_GLOBAL_KERNEL void my_dot_struct( GASQ float const * const a, GASQ float const * const b, GASQ float * const c, struct n_t const n ) {
  uint32_t const ix = GLOB_ID_1D;
  if( ix < n.n ) { c[ix] = a[ix] + b[ix]; }
}
"""


@pytest.mark.parametrize("func_name", ["my_dot", "my_dot_struct"])
def test_rtc_test_all_is_well(be, func_name):
    rtc = be.rtc
    data_sz = 10000
    rng = np.random.default_rng(1)
    a = rng.uniform(2.5, 7.5, data_sz).astype(np.float32); b = rng.uniform(2.5, 7.5, data_sz).astype(np.float32)
    c = np.full(data_sz, 123.456, np.float32)
    op = Op({"func_name": func_name}, {})
    rtc.compile([RtcFuncInfo(func_name, DOT_SRC, ["a", "b", "c", "n"], op)])
    for vn, v in (("a", a), ("b", b), ("c", c)):
        rtc.init_var_from_vect_float(vn, v)
    try:
        rfc = RtcFuncCall(func_name, {"a": RtcArg.var("a"), "b": RtcArg.var("b"), "c": RtcArg.var("c"), "n": RtcArg.scalar(data_sz, "uint32_t")},
                          tpb=256, blks=(data_sz + 255) // 256)
        cid = rtc.run(rfc)
        rtc.finish_and_sync()
        assert rtc.get_dur(cid, cid) > 0
        res = rtc.copy_var_to_nda("c")
        assert np.all(np.abs((a + b) - res) <= 1e-6)  # "All is Well."
        # missing arg -> rt_err; zero geometry -> rt_err (src/rtc_compute.cc:21-27)
        with pytest.raises(RtErr):
            rtc.run(RtcFuncCall(func_name, {"a": RtcArg.var("a")}, tpb=256, blks=1))
        with pytest.raises(RtErr):
            rtc.run(RtcFuncCall(func_name, rfc.arg_map, tpb=0, blks=0))
    finally:
        rtc.release_func(func_name)
        for vn in "abc":
            rtc.release_var(vn)
        rtc.release_per_call_id_data()


def test_var_semantics(be):
    rtc = be.rtc
    d = Dims.make("float", img=2, chan=3, y=4, x=5)
    rtc.create_var_with_dims("v", d)
    try:
        assert rtc.get_var_dims("v") == d
        assert not rtc.copy_var_to_nda("v").any()  # zero-filled on creation (src/nvrtc_util.cc:81-84)
        x = np.arange(120, dtype=np.float32).reshape(2, 3, 4, 5)
        rtc.copy_nda_to_var("v", x)
        flat = Dims.make("float", n=120)
        rtc.create_var_with_dims_as_reshaped_view_of_var("vv", flat, "v")
        assert np.array_equal(rtc.copy_var_to_nda("vv"), x.reshape(-1))  # a view shares the buffer
        rtc.set_var_to_zero("vv")
        assert not rtc.copy_var_to_nda("v").any()
        with pytest.raises(RtErr):
            rtc.create_var_with_dims("v", d)  # name must be new
        with pytest.raises(RtErr):
            rtc.create_var_with_dims_as_reshaped_view_of_var("bad", Dims.make("float", n=119), "v")
        with pytest.raises(RtErr):
            rtc.get_var_dims("nope")
        assert rtc.get_var_raw_native_pointer("v") == rtc.get_var_raw_native_pointer("vv") != 0
        rtc.release_var("v")
        assert np.array_equal(rtc.copy_var_to_nda("vv"), np.zeros(120, np.float32))  # view keeps the allocation alive
    finally:
        rtc.release_var("vv")


def test_gen_data_device_equals_oracle(be):
    rtc = be.rtc
    cases = [("sgemm", "a", Dims.make("float", K=37, M=52), lambda m: bo.gen_sgemm_a(37, 52, m)),
             ("sgemm", "b", Dims.make("float", K=37, N=44), lambda m: bo.gen_sgemm_b(37, 44, m)),
             ("Convolution", "in", Dims.make("float", img=2, chan=3, y=9, x=7), lambda m: bo.gen_conv_in(2, 3, 9, 7, m)),
             ("Convolution", "filts", Dims.make("float", out_chan=5, in_chan=3, y=3, x=3), lambda m: bo.gen_conv_filts(5, 3, 3, 3, m)),
             ("Convolution", "biases", Dims.make("float", out_chan=77), lambda m: bo.gen_conv_biases(77, m))]
    for t, an, dims, ref in cases:
        for mode in (2, 3, 4, 5, 600):
            rtc.create_var_with_dims("g", dims)
            try:
                rtc.run(gd.gen_call(t, an, "g", dims, mode, 0.25))
                got = rtc.copy_var_to_nda("g")
                want = ref(mode) + np.float32(0.25) if False else None
            finally:
                rtc.release_var("g")
            # oracle with the same vi
            if t == "sgemm" and an == "a": want = bo.gen_sgemm_a(37, 52, mode, 0.25)
            elif t == "sgemm": want = bo.gen_sgemm_b(37, 44, mode, 0.25)
            elif an == "in": want = bo.gen_conv_in(2, 3, 9, 7, mode, 0.25)
            elif an == "filts": want = bo.gen_conv_filts(5, 3, 3, 3, mode, 0.25)
            else: want = bo.gen_conv_biases(77, mode, 0.25)
            assert np.array_equal(got.reshape(-1), want.reshape(-1)), (t, an, mode)
    # batch-axis shards generate the matching slice of the global pattern (multi-GPU path, boda_amd/shard.py)
    d = Dims.make("float", K=37, M=20)
    rtc.create_var_with_dims("g", d)
    rtc.run(gd.gen_call("sgemm", "a", "g", d, 5, 0.0, shard_off=32, shard_glob=52))
    assert np.array_equal(rtc.copy_var_to_nda("g"), bo.gen_sgemm_a(37, 52, 5)[:, 32:52])
    rtc.release_var("g")
    d = Dims.make("float", img=2, chan=3, y=9, x=7)
    rtc.create_var_with_dims("g", d)
    rtc.run(gd.gen_call("Convolution", "in", "g", d, 5, 0.0, shard_off=3))
    assert np.array_equal(rtc.copy_var_to_nda("g"), bo.gen_conv_in(5, 3, 9, 7, 5)[3:5])
    rtc.release_var("g")
    rtc.release_per_call_id_data()


# ---------------------------------------------------------------------------------------------------------------
# SGEMM
# ---------------------------------------------------------------------------------------------------------------
def test_sgemm_golden_digests(be, golden_dir):
    for name, mode in (("sgemm-gen600", 600), ("sgemm-gen5", 5)):
        ow = read_wisdoms(os.path.join(golden_dir, "wisdom", name + ".wis"))[0]
        outs, prc = _run(be, ow.op, mode)
        vn, kg = ow.kgs[0]
        dg = Digest.from_array(outs[vn], kg.dims, kg.seed)
        assert kg.mrd_comp(dg, MRD) == "", name
        assert prc.rt_secs > 0


@pytest.mark.parametrize("M,N,K", [(64, 64, 64), (128, 128, 128), (384, 384, 384), (100, 36, 50), (33, 257, 19), (1, 1, 1),
                                   (260, 130, 70), (512, 64, 2048), (96, 1000, 363)])
def test_sgemm_vs_oracle_bit_exact(be, M, N, K):
    op = _sgemm_op(M, N, K)
    outs, _ = _run(be, op, 5, include_ins=True)
    want = bo.sgemm(outs["a"], outs["b"])
    assert np.array_equal(outs["a"], bo.gen_sgemm_a(K, M, 5))
    sd = SsdsDiff.of(want, outs["c"])
    assert sd.mrd < MRD, sd.basic_str()
    assert np.array_equal(want, outs["c"]), sd.basic_str()  # same fma chain -> same bits


@pytest.mark.parametrize("tile", ["64x64x16x1x1", "128x64x16x2x1", "64x128x16x1x2", "32x128x16x1x2", "96x128x16x1x2", "128x128x32x2x2", "256x128x16x4x2",
                                  "32x32x32x2x2x1x1x16", "64x64x64x4x4x1x1x16", "64x64x16x2x2x2x1x16", "64x64x16x2x2x2x1x32x2", "128x128x16x2x2x1x1x32x2",
                                  "128x128x16x2x2x2x1x32x1x1", "64x64x32x2x2x2x1x16x1x1",   # tenth field 1: as many staging waves as multiplying waves (round 4)
                                  "256x256x16x2x4x1x1x32x2",                                # the 256x256 tile = kernels/sgemm_big_f32.hip (ragged edges, K tail, fewer K tiles than its rounds)
                                  "128x128x8x3x4x2", "128x128x16x3x4x1", "256x128x8x3x4x1", "128x256x16x3x4x1",    # round 5: that kernel's 128 x 128 / 256 x 128 / 128 x 256 forms ("x3x4": its twelve waves)
                                  "64x64x16x2x2x4x1x32x2x3", "128x128x8x2x2x3x1x32x2x3", "64x128x16x2x2x4x1x32x2x3", "64x128x16x1x4x4x1x32x4x3", "128x64x16x2x2x4x1x32x2x3", "128x64x16x4x1x4x1x32x2x3",
                                  "64x256x16x1x8x2x1x32x2x3", "128x128x32x4x2x2x1x32x2x3"])   # round 6 (tenth field 3): the multiplying waves spelled out -- four of them, wave tiles of one row block
def test_sgemm_tiles_agree(be, tile):
    op = _sgemm_op(320, 448, 200)
    ref, _ = _run(be, op, 5)
    got, prc = _run(be, op, 5, tune=OpTune(hip_tile=tile))
    assert prc.launch["cfg"].startswith("x".join(tile.split("x")[:2]))
    stg = len(tile.split("x")) == 10 and tile.endswith("x3")
    assert (prc.launch["kernel"] == "bodahip_sgemm_big_f32") == (tile.startswith("256x256") or "x3x4" in tile or stg) and prc.launch["cfg"].endswith("_stg" if stg else "_sw") == (len(tile.split("x")) == 10)
    if stg: assert prc.launch["block"] == int(tile.split("x")[3]) * int(tile.split("x")[4]) * 64 + 256
    assert np.array_equal(ref["c"], got["c"])  # incl. the 16x16x4-MFMA tiles (suffix x16): same ascending-k fma chain


def test_sgemm_bad_tile_is_unsupported(be):
    with pytest.raises(UnsupErr):
        _run(be, _sgemm_op(64, 64, 64), 5, tune=OpTune(hip_tile="48x128x16x1x2"))


def test_sgemm_alias_cublas_and_mixed_types_unsupported(be):
    op = _sgemm_op(128, 256, 64)
    a, _ = _run(be, op, 5)
    b, prc = _run(be, op, 5, tune=OpTune(use_culibs=1))
    assert prc.op.get_func_name() == "cublas_sgemm" and np.array_equal(a["c"], b["c"])
    hop = parse_op("(str_vals=(type=sgemm),nda_vals=(a=(tn=half,dims=(K=64,M=64)),b=(dims=(K=64,N=64)),c=(tn=half,dims=(M=64,N=64))))")   # half a, float b
    with pytest.raises(UnsupErr):
        anno = add_codegen_annotations(hop, OpTune())
        profile_rcg_call(be, anno, None)
    dop = parse_op("(str_vals=(type=sgemm),nda_vals=(a=(tn=double,dims=(K=64,M=64)),b=(tn=double,dims=(K=64,N=64)),c=(tn=double,dims=(M=64,N=64))))")
    with pytest.raises(UnsupErr):
        profile_rcg_call(be, add_codegen_annotations(dop, OpTune()), None)


def test_sgemm_half_storage_fp32_math(be, golden_dir):
    """16-bit storage, fp32 math: the reference's reduced-precision precedent (sgemm with __tn__=half dims: vload_half -> float, fp32 fma chain,
    vstore_half; src/cnn_codegen.cc:440-449, gen_data through store_float_to_rp_half, test/rtc/gen_data_sgemm_a.cucl:20).  half -> float is exact and
    the fp32 MFMA chain is the reference's per-thread fmaf loop, so c is BIT-identical to the oracle run on the half-rounded operands and rounded
    to half (RNE) once.  The reference's own op (test/sgemm-ops-debug-half.txt: 2048^3) and ragged shapes (scalar 16-bit loads, tile edges)."""
    ops = read_ops(os.path.join(golden_dir, "ops", "sgemm-ops-debug-half.txt"))
    assert len(ops) == 1 and all(ops[0].get_dims(an).tn == "half" for an in "abc") and ops[0].sgemm_geom() == {"M": 2048, "N": 2048, "K": 2048}
    for op in ops + [parse_op(f"(str_vals=(type=sgemm),nda_vals=(a=(tn=half,dims=(K={K},M={M})),b=(tn=half,dims=(K={K},N={N})),c=(tn=half,dims=(M={M},N={N}))))")
                     for M, N, K in ((100, 36, 50), (33, 257, 19), (260, 130, 70), (1, 64, 64), (64, 3, 7))]:
        g = op.sgemm_geom()
        outs, prc = _run(be, op, 5, include_ins=True)
        assert prc.op.get_func_name() == "hip_sgemm" and prc.launch["kernel"] == "bodahip_sgemm_f16s" and outs["c"].dtype == np.float16
        a16 = bo.gen_sgemm_a(g["K"], g["M"]).astype(np.float16); b16 = bo.gen_sgemm_b(g["K"], g["N"]).astype(np.float16)
        assert np.array_equal(outs["a"], a16) and np.array_equal(outs["b"], b16)                      # the device generator rounds like numpy (RNE)
        want = bo.sgemm(a16.astype(np.float32), b16.astype(np.float32)).astype(np.float16)
        assert np.isfinite(want.astype(np.float32)).all()
        assert np.array_equal(outs["c"], want), (g, prc.launch["cfg"], SsdsDiff.of(want.astype(np.float32), outs["c"].astype(np.float32)).basic_str())
        assert prc.launch["algo_bytes"] == 2.0 * (g["K"] * g["M"] + g["K"] * g["N"] + g["M"] * g["N"])
    # mode 600 (a = 1000 m + k as half: exact below 2048, b = identity): c == a^T
    op = parse_op("(str_vals=(type=sgemm),nda_vals=(a=(tn=half,dims=(K=64,M=2)),b=(tn=half,dims=(K=64,N=64)),c=(tn=half,dims=(M=2,N=64))))")
    outs, _ = _run(be, op, 600, include_ins=True)
    assert np.array_equal(outs["c"], outs["a"].T)


def test_sgemm_full_sizes_exact_answer(be, golden_dir):
    """BASELINE config 2 sizes (test/sgemm-ops-full.txt): mode 600 => c[m,n] == 1000*m + n exactly (size-independent)."""
    ops = read_ops(os.path.join(golden_dir, "ops", "sgemm-ops-full.txt"))
    assert len(ops) == 17
    for op in ops:
        g = op.sgemm_geom()
        outs, prc = _run(be, op, 600)
        c = outs["c"]
        m = np.arange(g["M"], dtype=np.float32)[:, None] * np.float32(1000.0)
        n = np.arange(g["N"], dtype=np.float32)[None, :]
        assert np.array_equal(c, m + n), g
        assert prc.launch["flops"] == op.flops()


def test_sgemm_linearity_large(be):
    """c(vi=1 pattern) - c(vi=0 pattern): (a+1)^T(b+1) - a^T b == colsum(a)+rowsum... checked via small-K closed form."""
    op = _sgemm_op(1024, 768, 8)
    anno = add_codegen_annotations(op, OpTune())
    o0, _ = profile_rcg_call(be, anno, 5, 0.0, include_ins=True)
    o1, _ = profile_rcg_call(be, anno, 5, 1.0, include_ins=True)
    assert np.array_equal(o1["a"], o0["a"] + np.float32(1.0)) or SsdsDiff.of(o1["a"], o0["a"] + 1).mrd < 1e-6
    want = bo.sgemm(o1["a"], o1["b"])
    assert np.array_equal(want, o1["c"])


# ---------------------------------------------------------------------------------------------------------------
# Convolution
# ---------------------------------------------------------------------------------------------------------------
def _check_wisdom_file(be, golden_dir, name):
    ws = read_wisdoms(os.path.join(golden_dir, "wisdom", name + ".wis"))
    worst, n_exact = 0.0, 0
    for ow in ws:
        outs, _ = _run(be, ow.op, 5)
        vn, kg = ow.kgs[0]
        dg = Digest.from_array(outs[vn], kg.dims, kg.seed)
        res = kg.mrd_comp(dg, MRD)
        assert res == "", (ow.op.to_str(), res)
        worst = max(worst, kg.worst_scaled_rd(dg))
        n_exact += int(dg.to_hex() == kg.to_hex())
    return len(ws), worst, n_exact


def test_conv_golden_debug_and_gen5(be, golden_dir):
    assert _check_wisdom_file(be, golden_dir, "conv-debug")[0] == 2
    assert _check_wisdom_file(be, golden_dir, "conv-gen5")[0] == 1


def test_conv_golden_full_204(be, golden_dir):
    n, worst, n_exact = _check_wisdom_file(be, golden_dir, "conv-full-gen5")
    print(f"conv-full-gen5: {n} ops, worst scaled rel diff {worst:.3g}, bit-identical digests {n_exact}")
    assert n == 204 and worst < MRD


def test_conv_golden_3x3_42(be, golden_dir):
    n, worst, n_exact = _check_wisdom_file(be, golden_dir, "ops-prof-conv-3x3-cudnn-boda")
    print(f"3x3: {n} ops, worst scaled rel diff {worst:.3g}, bit-identical digests {n_exact}")
    assert n == 42 and worst < MRD


EDGE_CONVS = [  # B, C, H, W, OC, KH, KW, S, P   (incl. 1x1 stride 1/2 -> k1 path, 1x1 with padding -> general gather)
    (2, 19, 11, 11, 40, 1, 1, 1, 0), (3, 64, 14, 14, 128, 1, 1, 2, 0), (2, 8, 7, 7, 16, 1, 1, 1, 1),
    (1, 3, 12, 12, 16, 3, 3, 1, 1), (2, 5, 17, 13, 7, 5, 5, 2, 2), (3, 4, 9, 9, 33, 1, 1, 1, 0), (1, 8, 6, 6, 40, 6, 6, 1, 0),
    (2, 3, 35, 35, 96, 11, 11, 4, 0), (1, 16, 14, 14, 130, 7, 7, 2, 3), (5, 32, 7, 7, 64, 1, 1, 2, 0), (2, 6, 10, 10, 12, 3, 3, 1, 0),
    (1, 1, 5, 5, 1, 5, 5, 1, 2), (4, 20, 8, 8, 100, 3, 3, 1, 1),
    # multi-tile with padding: the first / last row windows of the tensor sit in different workgroups than the bulk
    (2, 3, 40, 40, 16, 7, 7, 2, 3), (3, 4, 33, 31, 20, 5, 5, 1, 2), (2, 3, 64, 64, 24, 11, 11, 4, 5), (1, 3, 150, 150, 64, 7, 7, 2, 3),
    # many-tap windows at a batch where the planner takes 256-pel tiles: K steps of 70 / 98 (two whole channels) -- the patch planner's own LDS bound applies, not the
    # BK x BJ image's (found by tools/fuzz_conv.py big in round 4: "unsupported tile configuration 32x256x70")
    (64, 2, 19, 54, 100, 7, 5, 1, 0), (64, 16, 42, 49, 24, 7, 7, 1, 0),
]


@pytest.mark.parametrize("shape", EDGE_CONVS)
def test_conv_vs_oracle_bit_exact(be, shape):
    op = _conv_op(*shape)
    outs, _ = _run(be, op, 5, include_ins=True)
    g = op.conv_geom()
    want = bo.conv_fwd(outs["in"], outs["filts"], outs["biases"], (g["SY"], g["SX"]), (g["PY"], g["PX"]), True)
    sd = SsdsDiff.of(want, outs["out"])
    assert sd.mrd < MRD, sd.basic_str()
    assert np.array_equal(want, outs["out"]), sd.basic_str()


# (B, C, pooled H, pooled W, OC, KH, KW, conv stride, conv pad, pool window, pool stride): the tensor read is the pooling's input, planes of (H - 1) * ps + pk
POOLED_CONVS = [(3, 5, 9, 9, 40, 3, 3, 1, 1, 3, 2), (2, 7, 13, 11, 33, 5, 5, 1, 2, 3, 2), (5, 6, 6, 6, 70, 3, 3, 1, 1, 2, 2), (2, 4, 12, 10, 20, 3, 3, 2, 1, 3, 1), (1, 3, 8, 8, 16, 2, 2, 1, 0, 3, 3),
                (7, 12, 6, 6, 130, 3, 3, 1, 1, 3, 2), (3, 96, 27, 27, 64, 5, 5, 1, 2, 3, 2)]


@pytest.mark.parametrize("shape", POOLED_CONVS, ids=lambda s: "x".join(str(v) for v in s))
@pytest.mark.parametrize("tile", ["", "64x256x16x1x4x2", "32x256x16x1x4x2", "128x128x16x2x2x2", "64x64x16x2x2x2"])
def test_conv_with_a_max_pooling_fused_in_front_bit_exact(be, shape, tile):
    """Round 5 (kernels/gemm_conv_f32.hip, PKH; cnn_op.fuse_f32_pool): hip_conv whose `in` is a POOLING's input -- every element of the convolution's LDS input patch is
    the maximum of its window, formed while the patch is staged.  Equal bit for bit to the oracle's max pooling followed by its convolution (the reference runs the two
    as two functions: test/rtc/pool.cucl, then the conv): windows 2 x 2 / 3 x 3, pool strides 1 / 2 / 3, convolutions with padding, stride 2 in y, tiles that straddle
    images, an odd channel count (K tail), every tile of the patch planner."""
    from boda_amd.cnn_op import f32_pool_fusable, fuse_f32_pool
    from boda_amd.op import Dims
    B, C, H, W, OC, KH, KW, S, P, pk, ps = shape
    if S != 1:   # (the patch form needs stride 1 in x: a strided convolution is not fusable, and says so)
        op = _conv_op(B, C, H, W, OC, KH, KW, S, P)
        assert not f32_pool_fusable(add_codegen_annotations(op, OpTune()), Dims(("img", "chan", "y", "x"), (B, C, (H - 1) * ps + pk, (W - 1) * ps + pk), "float"), (pk, pk), (ps, ps), (0, 0), 0)
        return
    op = _conv_op(B, C, H, W, OC, KH, KW, S, P)
    anno = add_codegen_annotations(op, OpTune(hip_tile=tile))
    uin = Dims(("img", "chan", "y", "x"), (B, C, (H - 1) * ps + pk, (W - 1) * ps + pk), "float")
    plain = add_codegen_annotations(op, OpTune())
    assert f32_pool_fusable(plain, uin, (pk, pk), (ps, ps), (0, 0), 0) and not f32_pool_fusable(plain, uin, (pk, pk), (ps, ps), (0, 0), 1) and not f32_pool_fusable(plain, uin, (pk, pk), (ps, ps), (1, 1), 0)
    fuse_f32_pool(anno, uin, (pk, pk), (ps, ps))
    assert anno.conv_geom()["H"] == H and anno.conv_geom()["UH"] == (H - 1) * ps + pk
    outs, prc = profile_rcg_call(be, anno, 5, 0.0, 1, include_ins=True, tile=tile)
    assert prc.launch["kernel"] == "bodahip_conv_f32"
    pooled = bo.pool_fwd(outs["in"], (pk, pk), (ps, ps), (0, 0), False)
    assert pooled.shape == (B, C, H, W)
    want = bo.conv_fwd(pooled, outs["filts"], outs["biases"], (S, S), (P, P), True)
    assert np.array_equal(want, outs["out"]), (prc.launch["cfg"], SsdsDiff.of(want, outs["out"]).basic_str())


KHO_CASES = [   # (shape, tile with an eleventh field = K segments per tile): every operand mode of the kernel, 32x32 and 16x16 MFMA tiles, register rings, more jobs than workgroups fit
    ((32, 64, 13, 13, 256, 3, 3, 1, 1), "64x64x18x2x2x2x1x32x1x0x3"),     # LDS input patch; 340 tiles x 3 segments on at most 512 resident workgroups
    ((16, 256, 13, 13, 128, 1, 1, 1, 0), "64x64x16x2x2x2x1x32x2x0x4"),    # 1x1 gather, two K tiles in flight
    ((8, 32, 27, 27, 96, 3, 3, 2, 0), "32x64x32x2x4x1x1x16x2x0x2"),       # table gather, 16x16x4 MFMA tiles
    ((4, 3, 67, 67, 96, 11, 11, 4, 0), "32x256x22x1x4x2x1x32x1x0x3"),     # row gather / row-decimated patch (conv1 shapes)
    ((24, 384, 6, 6, 1000, 3, 3, 1, 1), "128x128x36x2x2x2x1x32x1x0x5"),   # ragged out_chan tile row, tiles that straddle images
    ((40, 1024, 6, 6, 1000, 1, 1, 1, 0), "128x128x16x2x2x2x1x32x1x0x8"),  # NiN cccp8's shape: 8 x 12 tiles, eight segments
    ((2, 16, 9, 9, 40, 3, 3, 1, 1), "64x64x16x2x2x2x1x32x1x0x64"),        # more segments asked for than K steps: normalised down
]


@pytest.mark.parametrize("shape,tile", KHO_CASES, ids=lambda v: v if isinstance(v, str) else "x".join(map(str, v)))
def test_conv_sequential_k_hand_off_bit_exact(be, shape, tile):
    """Round 5 (kernels/gemm_conv_f32.hip -DKHO=1): a tile's K range runs as SEGMENTS on persistent workgroups that pull (tile, segment) jobs from one counter; segment s
    continues the fma chains of segment s - 1 from its stored accumulators.  Nothing is re-associated: the result equals the oracle (and the unsegmented kernel) bit for bit.
    Each case runs twice with DIFFERENT data (gen_data's `vi`) -- vars are freed and re-created at the same addresses, so the second run reuses the call's workspace: a stale
    slab, flag or counter would show --, and three times in a row on the same tensors (the workspace must be left clean by every launch)."""
    op = _conv_op(*shape)
    g = op.conv_geom()
    anno = add_codegen_annotations(op, OpTune(hip_tile=tile))
    for vi, iters in ((0.0, 1), (0.375, 3)):
        outs, prc = profile_rcg_call(be, anno, 5, vi, iters, include_ins=True, tile=tile)
        assert prc.launch["kernel"] == "bodahip_conv_f32" and "_h" in prc.launch["cfg"], prc.launch
        want = bo.conv_fwd(outs["in"], outs["filts"], outs["biases"], (g["SY"], g["SX"]), (g["PY"], g["PX"]), True)
        assert np.array_equal(want, outs["out"]), (vi, prc.launch, SsdsDiff.of(want, outs["out"]).basic_str())
    nkt = -(-(g["C"] * g["KH"] * g["KW"]) // int(prc.launch["cfg"].split("x")[2].split("_")[0]))
    assert int(prc.launch["cfg"].split("_h")[1]) <= nkt     # no empty segment


@pytest.mark.parametrize("tile", ["64x64x16x2x2x2x1x32x1x0x2", "128x128x16x2x2x2x1x32x1x0x3"])
def test_conv_k_hand_off_random_shapes_bit_exact(be, tile):
    """The fuzz sweep of test_conv_random_shapes_bit_exact with every launch cut into K segments (where the K loop has at least two steps)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from fuzz_conv import cases
    for sh in cases(24, 11 + len(tile)):
        op = _conv_op(*sh)
        outs, prc = _run(be, op, 5, tune=OpTune(hip_tile=tile), include_ins=True)
        g = op.conv_geom()
        want = bo.conv_fwd(outs["in"], outs["filts"], outs["biases"], (g["SY"], g["SX"]), (g["PY"], g["PX"]), True)
        assert np.array_equal(want, outs["out"]), (sh, prc.launch["cfg"])


@pytest.mark.parametrize("shape", [s for s in EDGE_CONVS if s[6] >= 2] + [(3, 24, 15, 15, 100, 3, 3, 1, 1), (2, 3, 35, 35, 96, 11, 11, 4, 0)])
def test_conv_row_gather_all_kernel_widths(be, shape, monkeypatch):
    """The row gather (J_MODE 6) is the default for KW >= 6 only; force it for every KW >= 2 (padding, strides, first-row and
    tensor-end windows, unaligned wide loads) and hold it to the same bit-exact bar."""
    monkeypatch.setenv("BODAHIP_ROW_GATHER_MIN_KW", "2")
    op = _conv_op(*shape)
    outs, _ = _run(be, op, 5, include_ins=True)
    g = op.conv_geom()
    cfg = be.rtc.last_launch()["cfg"]
    if not (g["OH"] == 1 and g["OW"] == 1 and g["PY"] == 0 and g["KH"] == shape[2]):  # ipconv shapes never gather
        assert "_m16" in cfg or int(cfg.split("x")[2].split("_")[0]) % g["KW"] == 0, cfg  # BK = rows * KW: the row gather was taken (32x32-MFMA tiles only)
    want = bo.conv_fwd(outs["in"], outs["filts"], outs["biases"], (g["SY"], g["SX"]), (g["PY"], g["PX"]), True)
    assert np.array_equal(want, outs["out"]), SsdsDiff.of(want, outs["out"]).basic_str()


@pytest.mark.parametrize("shape", [s for s in EDGE_CONVS if s[5] * s[6] >= 2] + [(3, 24, 15, 15, 100, 3, 3, 1, 1), (4, 96, 27, 27, 256, 5, 5, 1, 2)])
def test_conv_table_gather_when_patch_disabled(be, shape, monkeypatch):
    """Stride-1 KxK convs default to the LDS-patch kernel (J_MODE 7); the per-element table gather (J_MODE 2) behind it (still
    the default for strided 3x3/5x5 and padded 1x1) is held to the same bit-exact bar on the same shapes."""
    monkeypatch.setenv("BODAHIP_NO_PATCH", "1")
    monkeypatch.setenv("BODAHIP_NO_ROW_GATHER", "1")
    op = _conv_op(*shape)
    outs, _ = _run(be, op, 5, include_ins=True)
    g = op.conv_geom()
    want = bo.conv_fwd(outs["in"], outs["filts"], outs["biases"], (g["SY"], g["SX"]), (g["PY"], g["PX"]), True)
    assert np.array_equal(want, outs["out"]), SsdsDiff.of(want, outs["out"]).basic_str()


def test_conv_without_relu_and_alias(be):
    op = _conv_op(2, 6, 10, 10, 12, 3, 3, 1, 1)
    anno = add_codegen_annotations(op, OpTune(use_culibs=1))
    assert anno.get_func_name() == "cudnn_conv"
    anno.nda_vals["conv_has_relu"].v = (0,)
    outs, _ = profile_rcg_call(be, anno, 5, include_ins=True)
    want = bo.conv_fwd(outs["in"], outs["filts"], outs["biases"], (1, 1), (1, 1), False)
    assert want.min() < 0 and np.array_equal(want, outs["out"])


@pytest.mark.parametrize("tile", ["64x64x16x1x1", "128x64x16x2x1", "32x128x16x1x2", "96x128x16x1x2", "128x128x32x2x2", "128x256x16x2x4",
                                  "64x64x16x2x2x2x1x16", "32x64x32x2x2x1x1x16", "64x64x16x2x2x2x1x32x2", "128x128x32x2x2x1x1x32x2", "64x256x16x1x4x2",
                                  "64x256x16x1x4x2x1x32x1x1", "64x64x16x2x2x2x1x32x1x1",    # staging waves (LDS input patch staged by waves of their own)
                                  "128x128x16x2x4x2x1x32x2x2", "64x256x16x1x8x2x1x32x2x2", "96x256x16x1x8x1x1x32x2x2", "256x192x16x4x2x1x1x32x2x2", "128x512x8x2x4x1x1x32x4x2",
                                  "64x512x32x1x8x1x1x32x1x2", "128x128x16x2x2x1x1x32x2x2"])   # round 6: kernels/conv_big_f32.hip (tenth field 2): eight (four) multiplying + four staging waves
def test_conv_tiles_agree(be, tile):
    op = _conv_op(3, 24, 15, 15, 100, 3, 3, 1, 1)
    ref, _ = _run(be, op, 5)
    got, _ = _run(be, op, 5, tune=OpTune(hip_tile=tile))
    assert np.array_equal(ref["out"], got["out"])


CBIG_SHAPES = [(3, 24, 15, 15, 100, 3, 3, 1, 1), (2, 3, 35, 35, 96, 11, 11, 4, 0), (5, 96, 7, 7, 130, 1, 1, 1, 0), (4, 17, 9, 9, 70, 5, 5, 1, 2), (7, 16, 6, 6, 100, 6, 6, 1, 0),
               (3, 33, 13, 13, 33, 1, 1, 2, 0), (2, 10, 12, 12, 300, 3, 3, 2, 1), (9, 20, 1, 1, 50, 1, 1, 1, 0), (1, 4, 40, 40, 8, 3, 3, 1, 1), (40, 12, 13, 13, 70, 3, 3, 1, 1)]


@pytest.mark.parametrize("filt", ["scratch", "direct"])
@pytest.mark.parametrize("tile", ["256x256x16x2x4x1x1x32x2x2", "128x256x32x2x4x2x1x32x2x2", "96x512x8x1x8x1x1x32x2x2", "64x256x8x1x8x2x1x32x2x2", "192x256x16x2x4x1x1x32x1x2", "128x384x8x2x4x1x1x32x2x2"])
def test_conv_staging_wave_kernel_bit_exact(be, tile, filt, monkeypatch):
    """Round 6, kernels/conv_big_f32.hip (multiplying waves + staging waves): every pel form (LDS input patch read in place / 1x1 / table gather), both filter paths
    (k-major from the call's scratch behind bodahip_conv_big_xpose / straight from OIHW rows), wave tiles of 1-4 x 1-4 blocks incl. the pitch-four layouts of three blocks,
    one and two workgroups per CU, 3 and 4 LDS stages, K tails, ragged out_chans / pels, tiles that straddle images: each equals the oracle's fma chain bit for bit."""
    if filt == "direct": monkeypatch.setenv("BODAHIP_CBIG_IVW", "direct")
    for sh in CBIG_SHAPES:
        op = _conv_op(*sh)
        try:
            outs, prc = _run(be, op, 5, tune=OpTune(hip_tile=tile), include_ins=True)
        except UnsupErr:
            continue   # (a tile whose LDS stages do not fit this geometry)
        assert prc.launch["kernel"] == "bodahip_conv_big_f32", prc.launch
        want = bo.conv_fwd(outs["in"], outs["filts"], outs["biases"], (sh[7], sh[7]), (sh[8], sh[8]), True)
        assert np.array_equal(want, outs["out"]), (sh, prc.launch["cfg"])


def test_conv_staging_wave_kernel_is_the_plan_where_its_tiles_deal_out(be, monkeypatch):
    """The planner's rule (round 6): stride-1 multi-tap layers whose two-workgroups-per-CU tiles deal out over the CUs take the staging-wave kernel -- here forced onto
    small shapes (BODAHIP_CBIG=force) so that the DEFAULT path (no tile string) is what runs: equal to the oracle and to the round-3 kernel (BODAHIP_CBIG=off)."""
    for sh in [(40, 12, 13, 13, 70, 3, 3, 1, 1), (6, 20, 27, 27, 40, 5, 5, 1, 2), (3, 8, 9, 9, 260, 2, 2, 1, 0), (3, 3, 51, 51, 96, 11, 11, 4, 0), (2, 5, 40, 40, 100, 6, 6, 2, 0)]:   # (the last two: row-decimated patch)
        op = _conv_op(*sh)
        monkeypatch.setenv("BODAHIP_CBIG", "force")
        outs, prc = _run(be, op, 5, include_ins=True)
        assert prc.launch["kernel"] == "bodahip_conv_big_f32" and "_big" in prc.launch["cfg"], prc.launch
        want = bo.conv_fwd(outs["in"], outs["filts"], outs["biases"], (sh[7], sh[7]), (sh[8], sh[8]), True)
        assert np.array_equal(want, outs["out"]), (sh, prc.launch["cfg"])
        monkeypatch.setenv("BODAHIP_CBIG", "off")
        old, prc2 = _run(be, op, 5)
        assert prc2.launch["kernel"] == "bodahip_conv_f32" and np.array_equal(old["out"], outs["out"])


def test_conv_two_level_tiling_along_the_pels_bit_exact():
    """Round 6: a staging-wave plan whose tiles leave a mostly idle last round runs as a main launch over whole rounds of the CUs plus a tail launch of smaller tiles over the
    remaining pels (`pels<N+rest:` in the plan; by default only for launches of >= 50 GFLOP -- lowered here, in a process of its own because the threshold is read once).
    Same bits as the oracle and as the single launch (BODAHIP_CBIG_SPLIT=off)."""
    import subprocess, sys
    code = (
        "import os, sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "from boda_amd.cnn_op import OpTune, add_codegen_annotations\n"
        "from boda_amd.ops_prof import OpsBackend, profile_rcg_call\n"
        "from boda_amd.rtc import make_rtc, explain_plan\n"
        "from oracle import boda_oracle as bo\n"
        "from tools.cbig_probe import conv_op\n"
        "rtc = make_rtc('(be=hip)', 0); rtc.init(); be = OpsBackend(rtc)\n"
        "op = conv_op(160, 16, 13, 13, 384, 3, 3, 1, 1)\n"
        "anno = add_codegen_annotations(op, OpTune())\n"
        "plan = explain_plan(anno); assert 'pels<' in plan and '+rest:' in plan, plan\n"
        "outs, prc = profile_rcg_call(be, anno, 5, 0.0, 1, include_ins=True)\n"
        "want = bo.conv_fwd(outs['in'], outs['filts'], outs['biases'], (1, 1), (1, 1), True)\n"
        "assert np.array_equal(want, outs['out']), prc.launch\n"
        "os.environ['BODAHIP_CBIG_SPLIT'] = 'off'\n"
        "assert 'pels<' not in explain_plan(anno)\n"
        "one, _ = profile_rcg_call(be, anno, 5, 0.0, 1)\n"
        "assert np.array_equal(one['out'], outs['out'])\n"
        "print('OK', prc.launch['grid'])\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ); env["BODAHIP_CBIG_SPLIT_MIN_GFLOP"] = "0.1"
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])


@pytest.mark.parametrize("tile", ["128x128x16x2x2x2x4", "64x64x16x2x2x2x3", "96x128x16x1x2x2x7"])
def test_splitk_sgemm_and_conv_within_reference_tolerance(be, tile):
    op = _sgemm_op(300, 200, 1000)
    outs, prc = _run(be, op, 5, tune=OpTune(hip_tile=tile), include_ins=True)
    assert "_s" in prc.launch["cfg"]
    _assert_matches_oracle(bo.sgemm(outs["a"], outs["b"]), outs["c"], prc.launch)
    cop = _conv_op(3, 40, 9, 9, 70, 3, 3, 1, 1)
    outs, prc = _run(be, cop, 5, tune=OpTune(hip_tile=tile), include_ins=True)
    assert "_s" in prc.launch["cfg"]
    _assert_matches_oracle(bo.conv_fwd(outs["in"], outs["filts"], outs["biases"], (1, 1), (1, 1), True), outs["out"], prc.launch)


def test_ipconv_shaped_convs(be):
    """output 1x1 / no padding / kernel == input (the reference's ipconv case): plain GEMM path, vector and scalar K."""
    for shape in [(7, 16, 6, 6, 100, 6, 6, 1, 0), (5, 33, 1, 1, 50, 1, 1, 1, 0), (3, 3, 5, 5, 10, 5, 5, 1, 0)]:
        op = _conv_op(*shape)
        outs, prc = _run(be, op, 5, include_ins=True)
        g = op.conv_geom()
        want = bo.conv_fwd(outs["in"], outs["filts"], outs["biases"], (1, 1), (0, 0), True)
        _assert_matches_oracle(want, outs["out"], prc.launch)


def test_conv_reference_cucl_variants_are_reported_unsupported():
    op = _conv_op(1, 64, 14, 14, 64, 1, 1, 1, 0)
    with pytest.raises(UnsupErr):
        add_codegen_annotations(op, OpTune(k1conv=1))


ALEXNET_B256 = [  # BASELINE config 3: alexnet_ng_conv per-layer conv-ops at batch 256 (C,H,W,OC,K,S,P)
    (3, 227, 227, 96, 11, 4, 0), (96, 27, 27, 256, 5, 1, 2), (256, 13, 13, 384, 3, 1, 1), (384, 13, 13, 384, 3, 1, 1),
    (384, 13, 13, 256, 3, 1, 1), (256, 6, 6, 4096, 6, 1, 0), (4096, 1, 1, 4096, 1, 1, 0), (4096, 1, 1, 1000, 1, 1, 0)]


@pytest.mark.parametrize("layer", ALEXNET_B256)
def test_conv_alexnet_b256_batch_prefix_invariance(be, layer):
    """Full-size property: inputs are a hash of the flat index, so the first 2 images of the B=256 input ARE the B=2
    input; conv is per-image, so out(B=256)[:2] must equal the oracle's out(B=2) (bit-exact), and the last image's
    output must be finite and not all-zero (catches index overflow at full size)."""
    C, H, W, OC, K, S, P = layer
    op = _conv_op(256, C, H, W, OC, K, K, S, P)
    outs, prc = _run(be, op, 5)
    small = bo.run_op(_conv_op(2, C, H, W, OC, K, K, S, P), 5)
    _assert_matches_oracle(small["out"], outs["out"][:2], prc.launch)
    last = outs["out"][-1]
    assert np.isfinite(last).all() and last.max() > 0
    # spot-check the last image against the oracle too (own hash offsets)
    inp = bo.gen_conv_in(256, C, H, W, 5)[-1:]
    want = bo.conv_fwd(inp, small["filts"], small["biases"], (S, S), (P, P), True)
    _assert_matches_oracle(want[0], last, prc.launch)


def test_ops_prof_harness_end_to_end(be, golden_dir, tmp_path):
    """ops-prof protocol: kg tune + a second tune, digest check vs input wisdom, wisdom out; prints ***ALL IS WELL***."""
    import io
    from boda_amd.digest import write_wisdoms
    ops = read_ops(os.path.join(golden_dir, "ops", "conv-ops-debug-tmp.txt"))
    win = read_wisdoms(os.path.join(golden_dir, "wisdom", "conv-debug.wis"))
    buf = io.StringIO()
    wout, nfail, rows = ops_prof(be.rtc, ops, {"def": OpTune(), "t64": OpTune(hip_tile="64x64x16x1x1"), "ref-k1": OpTune(use_be="nvrtc", k1conv=1)},
                                 "def", 5, wisdom_in=win, write_runs=True, out=buf)
    txt = buf.getvalue()
    assert nfail == 0 and "***ALL IS WELL***" in txt
    assert "annotation failure" in txt  # the reference-CUCL tune is recorded as unsupported, not fatal
    assert [k[1].to_hex() for w in wout for k in w.kgs] != []  # digests written
    for w, wi in zip(wout, win):
        assert w.kgs[0][1].mrd_comp(wi.kgs[0][1], MRD) == ""
    p = tmp_path / "out.wis"
    write_wisdoms(str(p), wout)
    back = read_wisdoms(str(p))
    assert len(back) == 2 and all(any(r.be_plat_tag.startswith("hip:") for t in w.wisdoms for r in t.runs.values()) for w in back)


# ---------------------------------------------------------------------------------------------------------------
# bf16-operand kernels (BASELINE config 5).  The reference has no bf16 path: parity is UNPINNED for them by construction.
# Stated bounds: (1) vs the oracle fed the same bf16-rounded operands (only the fp32 summation order differs: the 16-deep
# MFMA is not a sequential chain, and the patch kernel orders k channel-innermost): max-rel-diff < 1e-3 (measured worst 5.2e-4 at
# K = 2400, AlexNet conv2's 5x5 x 96 channels, through conv_patch_bf16.hip; 2.3e-4 at K=2048 through the gather kernel);  (2) vs the exact fp32 oracle:
# normalised RMS error < 1e-2
# (bf16 has 8 mantissa bits: ~2^-9 relative rounding per operand).
# ---------------------------------------------------------------------------------------------------------------
MRD_BF16 = 1e-3


def _nrms(want, got):
    w = want.astype(np.float64); g = got.astype(np.float64)
    return float(np.sqrt(np.mean((w - g) ** 2)) / max(1e-30, np.sqrt(np.mean(w ** 2))))


@pytest.mark.parametrize("M,N,K", [(128, 128, 128), (100, 36, 50), (260, 130, 70), (512, 64, 2048), (33, 257, 19)])
def test_bf16_sgemm(be, M, N, K):
    outs, prc = _run(be, _sgemm_op(M, N, K), 5, tune=OpTune(hip_dtype="bf16"), include_ins=True)
    assert prc.op.get_func_name() == "hip_sgemm_bf16"
    want_b = bo.sgemm(bo.to_bf16(outs["a"]), bo.to_bf16(outs["b"]), f64acc=True)
    sd = SsdsDiff.of(want_b, outs["c"])
    assert not sd.has_nan() and sd.mrd < MRD_BF16, sd.basic_str()
    assert _nrms(bo.sgemm(outs["a"], outs["b"], f64acc=True), outs["c"]) < 1e-2


@pytest.mark.parametrize("shape", EDGE_CONVS + [(4, 96, 27, 27, 256, 5, 5, 1, 2), (3, 256, 13, 13, 384, 3, 3, 1, 1), (8, 256, 6, 6, 512, 6, 6, 1, 0)])
def test_bf16_conv(be, shape):
    op = _conv_op(*shape)
    outs, prc = _run(be, op, 5, tune=OpTune(hip_dtype="bf16"), include_ins=True)
    assert prc.op.get_func_name() == "hip_conv_bf16"
    g = op.conv_geom()
    want_b = bo.conv_fwd(bo.to_bf16(outs["in"]), bo.to_bf16(outs["filts"]), outs["biases"], (g["SY"], g["SX"]), (g["PY"], g["PX"]), True)
    sd = SsdsDiff.of(want_b, outs["out"])
    assert not sd.has_nan() and sd.mrd < MRD_BF16, sd.basic_str()
    want = bo.conv_fwd(outs["in"], outs["filts"], outs["biases"], (g["SY"], g["SX"]), (g["PY"], g["PX"]), True)
    assert _nrms(want, outs["out"]) < 1e-2


@pytest.mark.parametrize("net", ["googlenet_conv", "resnet-50"])
def test_every_conv_layer_of_config5_nets_bit_exact(be, net):
    """BASELINE config 5's layer lists (64 GoogLeNet convs, 53 ResNet-50 convs + fc; shapes from the prototxts via our own reader):
    every distinct layer at batch 2 through its default plan -- patch / row-gather / 1x1 / table modes, strided and padded,
    first and last tiles of multi-tile grids -- must equal the oracle bit for bit."""
    import bench
    seen = set()
    for op in bench.net_conv_ops(net, 2):
        key = op.to_str()
        if key in seen:
            continue
        seen.add(key)
        outs, prc = _run(be, op, 5, include_ins=True)
        g = op.conv_geom()
        want = bo.conv_fwd(outs["in"], outs["filts"], outs["biases"], (g["SY"], g["SX"]), (g["PY"], g["PX"]), True)
        assert np.array_equal(want, outs["out"]), (key, prc.launch["cfg"], SsdsDiff.of(want, outs["out"]).basic_str())
    assert len(seen) >= 20


def test_reference_shaped_cucl_sgemm_matches_oracle(be):
    """boda_amd/ref_style.py: a register-tiled 8x8-per-thread sgemm in CUCL dialect (the structure of the reference's default
    sgemm variant) through the generic hiprtc path -- same ascending-k fma chain, so the same bits as the oracle and the MFMA kernel."""
    from boda_amd import ref_style
    rtc = be.rtc
    ref_style.compile_into(rtc)
    M, N, K = 192, 256, 104
    rng = np.random.default_rng(5)
    a = rng.uniform(-5, 5, (K, M)).astype(np.float32); b = rng.uniform(-5, 5, (K, N)).astype(np.float32)
    for vn, arr, d in (("rs_a", a, Dims.make("float", K=K, M=M)), ("rs_b", b, Dims.make("float", K=K, N=N)), ("rs_c", None, Dims.make("float", M=M, N=N))):
        rtc.create_var_with_dims(vn, d)
        if arr is not None:
            rtc.copy_nda_to_var(vn, arr)
    try:
        rtc.run(ref_style.call("rs_a", "rs_b", "rs_c", M, N, K)); rtc.finish_and_sync()
        got = rtc.copy_var_to_nda("rs_c")
        want = bo.sgemm(a, b)
        assert SsdsDiff.of(want, got).mrd < MRD
        assert np.array_equal(want, got)
        with pytest.raises(UnsupErr):
            ref_style.call("rs_a", "rs_b", "rs_c", 100, 256, 104)
    finally:
        for vn in ("rs_a", "rs_b", "rs_c"):
            rtc.release_var(vn)
        rtc.release_per_call_id_data()


def test_cucl_template_instance_runs_through_generic_path(be):
    """A CUCL *template* (magic-comment arg decls, %(...) variables, CUCL IX index expressions) instantiated by
    boda_amd/cucl_template.py -- the reference's rtc_func_gen layer -- compiled and launched by the backend with the generated
    arg list and geometry."""
    from boda_amd.cucl_template import instantiate, parse_template
    from boda_amd.op import Nda
    from test_cucl_template_cpu import OWN
    rtc = be.rtc
    t = parse_template("own", OWN)
    din, dout = Dims.make("float", img=2, chan=3, y=9, x=7), Dims.make("float", img=2, chan=3, y=5, x=4)
    op = Op({"type": "own", "func_name": "own"}, {"in": Nda(din), "out": Nda(dout), "stride": Nda(Dims(("y", "x"), (2, 2), "none"), "none"),
                                                  "shift": Nda(None, "uint32_t", (7,))})
    inst = instantiate(t, op, "own__gpu0")
    rtc.compile([RtcFuncInfo(inst.func_name, inst.src, inst.arg_names, op)])
    x = np.arange(din.dims_prod(), dtype=np.float32).reshape(din.sizes)
    rtc.create_var_with_dims("tpl_in", din); rtc.create_var_with_dims("tpl_out", dout); rtc.copy_nda_to_var("tpl_in", x)
    try:
        am = {"in": RtcArg.var("tpl_in"), "out": RtcArg.var("tpl_out"), "stride": RtcArg.ref(op.get_dims("stride")), "shift": RtcArg.scalar(7, "uint32_t")}
        rtc.run(RtcFuncCall(inst.func_name, am, tpb=inst.tpb, blks=inst.blks)); rtc.finish_and_sync()
        assert np.array_equal(rtc.copy_var_to_nda("tpl_out"), x[:, :, ::2, ::2][:, :, :5, :4] + 7)
    finally:
        rtc.release_var("tpl_in"); rtc.release_var("tpl_out"); rtc.release_func(inst.func_name); rtc.release_per_call_id_data()


def test_conv_writes_channel_slice_of_wider_output(be):
    """hip_conv's optional out_chan_off: `out` is a wider tensor, the conv writes exactly its channel range of it (Concat
    elimination in the full-net driver) -- bit-exact values, every other channel untouched, bad offsets refused."""
    rtc = be.rtc
    op = _conv_op(3, 6, 10, 10, 12, 3, 3, 1, 1)
    anno = add_codegen_annotations(op, OpTune())
    fn = anno.get_func_name()
    rtc.compile([RtcFuncInfo("slice_conv", "", [a for a, _ in NATIVE_ARGS[fn]], anno)])
    g = op.conv_geom()
    x = bo.gen_conv_in(3, 6, 10, 10); f = bo.gen_conv_filts(12, 6, 3, 3); b = bo.gen_conv_biases(12)
    wide = Dims.make("float", img=3, chan=30, y=10, x=10)
    names = {"in": ("sl_in", anno.get_dims("in"), x), "filts": ("sl_f", anno.get_dims("filts"), f), "biases": ("sl_b", anno.get_dims("biases"), b),
             "out": ("sl_out", wide, np.full(wide.sizes, 7.0, np.float32))}
    for vn, d, arr in names.values():
        rtc.create_var_with_dims(vn, d); rtc.copy_nda_to_var(vn, arr)
    try:
        am = {an: RtcArg.var(names[an][0]) for an in names}
        am["stride"] = RtcArg.ref(anno.get_dims("stride")); am["in_pad"] = RtcArg.ref(anno.get_dims("in_pad"))
        am["out_chan_off"] = RtcArg.scalar(11, "uint32_t")
        rtc.run(RtcFuncCall("slice_conv", am)); rtc.finish_and_sync()
        got = rtc.copy_var_to_nda("sl_out")
        want = bo.conv_fwd(x, f, b, (1, 1), (1, 1), True)
        assert np.array_equal(got[:, 11:23], want) and (got[:, :11] == 7).all() and (got[:, 23:] == 7).all()
        am["out_chan_off"] = RtcArg.scalar(19, "uint32_t")     # 19 + 12 > 30
        with pytest.raises(RtErr):
            rtc.run(RtcFuncCall("slice_conv", am))
    finally:
        for vn, _, _ in names.values():
            rtc.release_var(vn)
        rtc.release_func("slice_conv"); rtc.release_per_call_id_data()


@pytest.mark.parametrize("tile", ["", "64x256x32x1x4x2", "128x128x16x2x2x2", "64x64x32x2x2x1x1x32x4",    # (the fourth: a ring of four register-staged K tiles, every operand mode)
                                  "128x128x16x2x4x2x1x32x2x2", "96x256x16x1x8x1x1x32x2x2"])              # round 6: the staging-wave kernel (patch / 1x1 / gather form as the shape allows)
def test_conv_random_shapes_bit_exact(be, tile):
    """Seeded random-shape sweep (tools/fuzz_conv.py; kernel sizes 1..11, strides 1..4, paddings, ragged channel counts): whatever
    operand mode the planner picks, with the default and with two forced workgroup tiles, equals the oracle bit for bit.
    (The tool was run over 1350 cases across five tiles when the modes were written: no mismatch.)"""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from fuzz_conv import cases
    for sh in cases(30, 5 + len(tile)):
        op = _conv_op(*sh)
        outs, prc = _run(be, op, 5, tune=OpTune(hip_tile=tile), include_ins=True)
        g = op.conv_geom()
        want = bo.conv_fwd(outs["in"], outs["filts"], outs["biases"], (g["SY"], g["SX"]), (g["PY"], g["PX"]), True)
        assert np.array_equal(want, outs["out"]), (sh, prc.launch["cfg"])


@pytest.mark.parametrize("tile", ["", "128x128x16x2x2x2", "256x256x16x4x4x1", "32x32x64x2x2x1x1x16x2", "32x32x32x2x2x1x1x16x8", "64x64x32x2x2x1x1x32x4", "64x64x16x2x2x1x1x32x6"])   # (the last three: rings of 8 / 4 / 6 register-staged K tiles)
def test_sgemm_random_shapes_bit_exact(be, tile):
    """Seeded random (M, N, K) incl. sizes that are not multiples of 4 / of the tile: default plan and three forced tiles == oracle."""
    rng = np.random.default_rng(3 + len(tile))
    for _ in range(12):
        M, N, K = (int(rng.integers(1, 700)) for _ in range(3))
        if rng.random() < 0.4:
            M, N = 4 * (M // 4 + 1), 4 * (N // 4 + 1)      # the float4 loader modes
        op = _sgemm_op(M, N, K)
        outs, prc = _run(be, op, 5, tune=OpTune(hip_tile=tile), include_ins=True)
        want = bo.sgemm(outs["a"], outs["b"])
        assert np.array_equal(want, outs["c"]), ((M, N, K), prc.launch["cfg"])


@pytest.mark.parametrize("shape", [(4540, 4096, 520), (4100, 4096, 516), (4348, 4100, 600), (6144, 6144, 512)], ids=lambda s: "x".join(map(str, s)))
def test_sgemm_two_level_tiling_bit_exact(be, shape):
    """The split launch (256x256 tiles over whole rounds of tile rows + small tiles over the remaining rows) == the unsplit launch == oracle,
    on shapes where the remainder is not a multiple of any tile (M = 4540: 17 full tile rows of 16 -> 16 rows main, 444 rows tail), is a single
    4-row sliver (4100), meets a ragged last tile column (N = 4100), and on one of the benchmark's own sizes (6144^2, K cut to 512)."""
    M, N, K = shape
    op = _sgemm_op(M, N, K)
    outs, prc = _run(be, op, 5, include_ins=True)
    assert prc.launch["cfg"].startswith("256x256") and prc.launch["grid"] > 256, prc.launch     # main tiles + tail tiles
    os.environ["BODAHIP_NO_SGEMM_SPLIT"] = "1"
    try: outs1, prc1 = _run(be, op, 5)
    finally: del os.environ["BODAHIP_NO_SGEMM_SPLIT"]
    assert np.array_equal(outs["c"], outs1["c"])
    assert np.array_equal(bo.sgemm(outs["a"], outs["b"]), outs["c"])


@pytest.mark.parametrize("shape,grid,oracle", [((5120, 5120, 768), 16 * 80 + 16 * 32 + 64 * 16, True), ((10240, 10240, 512), 32 * 80 + 8 * 64 + 32 * 32, False), ((4096, 10240, 520), 16 * 64 + 64 * 32, True)],
                         ids=lambda s: "x".join(map(str, s)) if isinstance(s, tuple) else str(s))
def test_sgemm_guillotine_parts_bit_exact(be, shape, grid, oracle):
    """Round 6: the two-level tiling generalised -- c cut into rectangles (rows | columns), each one launch of one tile form in whole rounds of the CUs (256 x 128 tiles: two
    per CU -> rounds of 512; 64 x 64: four per CU): 10240^2 = rows < 8192 on 256 x 128 (2560 tiles) + the last rows' first 8192 columns (512) + a 2048^2 corner on 64 x 64
    (1024); 5120^2 = a 1024-row strip on 64 x 64 + 4096^2 on 256 x 128 (512) + a 4096 x 1024 strip.  Every output is one launch's one ascending-k chain: equal to the
    single launch bit for bit, and to the oracle."""
    M, N, K = shape
    op = _sgemm_op(M, N, K)
    outs, prc = _run(be, op, 5, include_ins=True)
    assert prc.launch["kernel"] == "bodahip_sgemm_big_f32" and prc.launch["grid"] == grid, prc.launch
    os.environ["BODAHIP_NO_SGEMM_PARTS"] = "1"; os.environ["BODAHIP_NO_SGEMM_SPLIT"] = "1"
    try: outs1, prc1 = _run(be, op, 5)
    finally: del os.environ["BODAHIP_NO_SGEMM_PARTS"]; del os.environ["BODAHIP_NO_SGEMM_SPLIT"]
    assert prc1.launch["grid"] != grid and np.array_equal(outs["c"], outs1["c"])
    if oracle: assert np.array_equal(bo.sgemm(outs["a"], outs["b"]), outs["c"])


def test_cucl_template_dyn_dims_per_call(be):
    """A template with an OUT_DYN argument: one generated function serves any dims; the cai__* arguments and the launch geometry
    come from Instance.call_args per call (the reference's rcg_func_call_t::run flow)."""
    from boda_amd.cucl_template import instantiate, parse_template
    from boda_amd.op import Nda
    from test_cucl_template_cpu import DYN
    rtc = be.rtc
    op = Op({"type": "dyn", "func_name": "dyn"}, {"a": Nda(Dims.make("float", K=0, M=0)), "vi": Nda(None, "float", None)})
    inst = instantiate(parse_template("dyn", DYN), op, "dyn__gpu0")
    rtc.compile([RtcFuncInfo(inst.func_name, inst.src, inst.arg_names, op)])
    try:
        for K, M in ((37, 20), (5, 301)):
            d = Dims.make("float", K=K, M=M)
            vn = f"dyn_a_{K}"; rtc.create_var_with_dims(vn, d)
            vals, tpb, blks = inst.call_args({"a": d})
            am = {"a": RtcArg.var(vn), "vi": RtcArg.scalar(0.5, "float")}
            am.update({k: RtcArg.scalar(v, "int32_t") for k, v in vals.items()})
            rtc.run(RtcFuncCall(inst.func_name, am, tpb=tpb, blks=blks)); rtc.finish_and_sync()
            kk, mm = np.meshgrid(np.arange(K), np.arange(M), indexing="ij")
            assert np.array_equal(rtc.copy_var_to_nda(vn), (0.5 + kk * 1000 + mm + M).astype(np.float32))
            rtc.release_var(vn)
    finally:
        rtc.release_func(inst.func_name); rtc.release_per_call_id_data()


@pytest.mark.parametrize("shape", [(2, 6, 10, 10, 12, 3, 3, 1, 1), (3, 5, 17, 13, 70, 5, 5, 2, 2), (2, 19, 11, 11, 40, 1, 1, 1, 0), (2, 3, 35, 35, 96, 11, 11, 4, 0)])
def test_reference_shaped_cucl_conv_matches_oracle(be, shape):
    """boda_amd/ref_style.py: the register-tiled generic-conv structure of the reference as CUCL source through the generic path --
    a second GPU implementation of Convolution+bias+ReLU; same ascending-k fma chain, so the same bits as the oracle."""
    from boda_amd import ref_style
    rtc = be.rtc
    ref_style.compile_into(rtc)
    op = _conv_op(*shape); g = op.conv_geom()
    x = bo.gen_conv_in(*shape[:4]); f = bo.gen_conv_filts(shape[4], shape[1], shape[5], shape[6]); b = bo.gen_conv_biases(shape[4])
    names = {"in": ("rsc_in", x), "filts": ("rsc_f", f), "biases": ("rsc_b", b), "out": ("rsc_out", None)}
    for an, (vn, arr) in names.items():
        rtc.create_var_with_dims(vn, op.get_dims(an))
        if arr is not None:
            rtc.copy_nda_to_var(vn, arr)
    try:
        rtc.run(ref_style.conv_call("rsc_f", "rsc_b", "rsc_in", "rsc_out", g)); rtc.finish_and_sync()
        want = bo.conv_fwd(x, f, b, (g["SY"], g["SX"]), (g["PY"], g["PX"]), True)
        assert np.array_equal(want, rtc.copy_var_to_nda("rsc_out"))
    finally:
        for vn, _ in names.values():
            rtc.release_var(vn)
        rtc.release_per_call_id_data()


K1S_CASES = [  # (B, C, H, W, OC, spec): odd / tiny in_chan counts, out_chans that are not a multiple of the wave's rows, ragged pel tails,
    (3, 96, 11, 9, 96, "1x4x3x1"),          # blocks straddling images, several out_chan tiles
    (2, 7, 5, 5, 40, "1x2x2x1"),
    (5, 33, 13, 13, 70, "2x2x2x2"),
    (1, 2, 1, 1, 3, "1x1x1x1"),
    (4, 64, 14, 14, 256, "8x1x1x2"),
    (2, 128, 8, 8, 130, "4x2x1x1"),
    (2, 1, 9, 7, 33, "1x4x2x2"),
    (6, 24, 20, 20, 64, "1x8x2x1"),
    # kernels/k1_quad_f32.hip ("qWJxOCBxRING": 128-pel blocks of one image, 16 bytes per lane): planes shorter than a block, tail blocks whose surplus
    (3, 96, 11, 9, 96, "q4x3x8"),           # columns clamp to the last four pels, the smallest plane (4 pels), odd in_chan counts, ragged out_chans, two out_chan
    (2, 7, 5, 5, 40, "q2x2x4"),             # tiles, rings of one step and of every step, more workgroup slots than blocks and fewer
    (5, 33, 13, 13, 70, "q4x3x1"),
    (1, 2, 2, 2, 3, "q1x1x1"),
    (2, 64, 23, 23, 130, "q4x3x8"),
    (6, 24, 20, 20, 64, "q8x2x6"),
    (9, 96, 55, 55, 96, "q4x3x8"),
    (3, 32, 16, 16, 32, "q2x1x16"),
]


@pytest.mark.parametrize("case", K1S_CASES, ids=lambda c: "x".join(str(v) for v in c[:5]) + "_" + c[5])
@pytest.mark.parametrize("relu", [True, False])
def test_k1_stream_kernel_bit_exact(be, case, relu):
    """The streaming 1x1 kernel (kernels/k1_stream_f32.hip, tune key k1_stream) forced onto small shapes: same bits as the oracle,
    and a guard band around the output stays untouched (its stores rely on the buffer range check for the pel tail)."""
    rtc = be.rtc
    B, C, H, W, OC, spec = case
    op = _conv_op(B, C, H, W, OC, 1, 1, 1, 0)
    anno = add_codegen_annotations(op, OpTune())
    anno.nda_vals["conv_has_relu"].v = (int(relu),)
    fn = anno.get_func_name()
    rtc.compile([RtcFuncInfo("k1s_conv", "", [a for a, _ in NATIVE_ARGS[fn]], anno)])
    x = bo.gen_conv_in(B, C, H, W); f = bo.gen_conv_filts(OC, C, 1, 1); b = bo.gen_conv_biases(OC)
    wide = Dims.make("float", img=B, chan=OC + 5, y=H, x=W)
    names = {"in": ("k1_in", anno.get_dims("in"), x), "filts": ("k1_f", anno.get_dims("filts"), f), "biases": ("k1_b", anno.get_dims("biases"), b),
             "out": ("k1_out", wide, np.full(wide.sizes, 7.0, np.float32))}
    for vn, d, arr in names.values():
        rtc.create_var_with_dims(vn, d); rtc.copy_nda_to_var(vn, arr)
    try:
        am = {an: RtcArg.var(names[an][0]) for an in names}
        am["stride"] = RtcArg.ref(anno.get_dims("stride")); am["in_pad"] = RtcArg.ref(anno.get_dims("in_pad"))
        am["out_chan_off"] = RtcArg.scalar(2, "uint32_t")
        rtc.set_tune("k1_stream", spec)
        rtc.run(RtcFuncCall("k1s_conv", am)); rtc.finish_and_sync()
        assert rtc.last_launch()["kernel"] == ("bodahip_k1_quad_f32" if spec.startswith("q") else "bodahip_k1_stream_f32")
        got = rtc.copy_var_to_nda("k1_out")
        want = bo.conv_fwd(x, f, b, (1, 1), (0, 0), relu)
        assert np.array_equal(got[:, 2:2 + OC], want), SsdsDiff.of(want, got[:, 2:2 + OC]).basic_str()
        assert (got[:, :2] == 7).all() and (got[:, 2 + OC:] == 7).all()
        rtc.set_tune("k1_stream", "off")
        rtc.run(RtcFuncCall("k1s_conv", am)); rtc.finish_and_sync()
        assert rtc.last_launch()["kernel"] == "bodahip_conv_f32" and np.array_equal(rtc.copy_var_to_nda("k1_out"), got)
    finally:
        rtc.set_tune("k1_stream", "")
        for vn, _, _ in names.values():
            rtc.release_var(vn)
        rtc.release_func("k1s_conv"); rtc.release_per_call_id_data()


K1_CHAIN_CASES = [  # (B, C, H, W, MID, OC2): odd in_chan / intermediate / out_chan counts (padded K steps of both convolutions, ragged row blocks), planes shorter than a
    (2, 5, 9, 9, 7, 13),            # 128-pel block and planes with tail blocks, more blocks than workgroup slots, one to three row blocks either side, 128 out_chans
    (3, 96, 11, 9, 96, 96),
    (2, 33, 20, 20, 64, 100),
    (1, 2, 2, 2, 1, 3),
    (5, 48, 13, 13, 33, 128),
    (9, 96, 55, 55, 96, 96),
    (2, 128, 23, 23, 95, 32),
]


@pytest.mark.parametrize("case", K1_CHAIN_CASES, ids=lambda c: "x".join(str(v) for v in c))
@pytest.mark.parametrize("relus", [(1, 1), (0, 1), (1, 0)], ids=["relu_relu", "lin_relu", "relu_lin"])
def test_k1_chain_bit_exact(be, case, relus):
    """hip_conv_k1_chain (kernels/k1_quad_f32.hip -DCHAIN=1): two 1x1 convolutions as one launch, the intermediate tensor in the accumulator registers -- the same bits
    as the oracle's two convolutions run one after the other (the intermediate rounded to fp32, bias and ReLU in between), into a channel slice of a wider tensor with
    an untouched guard band; with `mid` given, the intermediate tensor too."""
    from boda_amd.cnn_op import annotate_k1_chain, k1_chain_applies
    rtc = be.rtc
    B, C, H, W, MID, OC2 = case
    a = add_codegen_annotations(_conv_op(B, C, H, W, MID, 1, 1, 1, 0), OpTune()); b = add_codegen_annotations(_conv_op(B, MID, H, W, OC2, 1, 1, 1, 0), OpTune())
    assert k1_chain_applies(a, b)
    ch = annotate_k1_chain(a, b, *relus)
    args = [x for x, _ in NATIVE_ARGS["hip_conv_k1_chain"]]
    rtc.compile([RtcFuncInfo("k1c", "", args, ch), RtcFuncInfo("k1c_mid", "", args + ["mid"], ch)])
    x = bo.gen_conv_in(B, C, H, W); f1 = bo.gen_conv_filts(MID, C, 1, 1); b1 = bo.gen_conv_biases(MID)
    f2 = (bo.gen_conv_filts(OC2, MID, 1, 1) * np.float32(0.25)).astype(np.float32); b2 = bo.gen_conv_biases(OC2)
    wide = Dims.make("float", img=B, chan=OC2 + 5, y=H, x=W)
    names = {"in": ("k1c_in", a.get_dims("in"), x), "filts": ("k1c_f1", a.get_dims("filts"), f1), "biases": ("k1c_b1", a.get_dims("biases"), b1),
             "filts2": ("k1c_f2", b.get_dims("filts"), f2), "biases2": ("k1c_b2", b.get_dims("biases"), b2), "out": ("k1c_out", wide, np.full(wide.sizes, 7.0, np.float32)),
             "mid": ("k1c_mid_v", a.get_dims("out"), np.full(a.get_dims("out").sizes, 3.0, np.float32))}
    for vn, d, arr in names.values():
        rtc.create_var_with_dims(vn, d); rtc.copy_nda_to_var(vn, arr)
    try:
        am = {an: RtcArg.var(names[an][0]) for an in names if an != "mid"}
        am["stride"] = RtcArg.ref(a.get_dims("stride")); am["in_pad"] = RtcArg.ref(a.get_dims("in_pad")); am["out_chan_off"] = RtcArg.scalar(2, "uint32_t")
        mid_w = bo.conv_fwd(x, f1, b1, (1, 1), (0, 0), bool(relus[0])); want = bo.conv_fwd(mid_w, f2, b2, (1, 1), (0, 0), bool(relus[1]))
        rtc.run(RtcFuncCall("k1c", am)); rtc.finish_and_sync()
        assert rtc.last_launch()["kernel"] == "bodahip_k1_chain_f32"
        got = rtc.copy_var_to_nda("k1c_out")
        assert np.array_equal(got[:, 2:2 + OC2], want), SsdsDiff.of(want, got[:, 2:2 + OC2]).basic_str()
        assert (got[:, :2] == 7).all() and (got[:, 2 + OC2:] == 7).all()
        assert (rtc.copy_var_to_nda("k1c_mid_v") == 3).all()       # (not asked for: not written)
        rtc.copy_nda_to_var("k1c_out", np.full(wide.sizes, 7.0, np.float32))
        rtc.run(RtcFuncCall("k1c_mid", dict(am, mid=RtcArg.var("k1c_mid_v")))); rtc.finish_and_sync()
        got = rtc.copy_var_to_nda("k1c_out")
        assert np.array_equal(got[:, 2:2 + OC2], want) and (got[:, :2] == 7).all() and (got[:, 2 + OC2:] == 7).all()
        assert np.array_equal(rtc.copy_var_to_nda("k1c_mid_v"), mid_w)
    finally:
        for vn, _, _ in names.values():
            rtc.release_var(vn)
        rtc.release_func("k1c"); rtc.release_func("k1c_mid"); rtc.release_per_call_id_data()


def test_k1_chain_refuses_what_it_does_not_cover(be):
    """More than 96 intermediate channels, a padded or strided member: the annotation refuses (UnsupErr), nothing is launched."""
    from boda_amd.cnn_op import annotate_k1_chain, k1_chain_applies
    from boda_amd.op import UnsupErr
    mk = lambda *a_: add_codegen_annotations(_conv_op(*a_), OpTune())
    assert not k1_chain_applies(mk(2, 8, 9, 9, 97, 1, 1, 1, 0), mk(2, 97, 9, 9, 8, 1, 1, 1, 0))
    assert not k1_chain_applies(mk(2, 8, 9, 9, 16, 1, 1, 1, 0), mk(2, 16, 9, 9, 129, 1, 1, 1, 0))
    assert not k1_chain_applies(mk(2, 8, 9, 9, 16, 3, 3, 1, 1), mk(2, 16, 9, 9, 8, 1, 1, 1, 0))
    assert not k1_chain_applies(mk(2, 8, 9, 9, 16, 1, 1, 1, 0), mk(2, 16, 9, 9, 8, 3, 3, 1, 1))
    assert not k1_chain_applies(mk(2, 8, 9, 9, 16, 1, 1, 1, 0), mk(2, 24, 9, 9, 8, 1, 1, 1, 0))
    with pytest.raises(UnsupErr):
        annotate_k1_chain(mk(2, 8, 9, 9, 97, 1, 1, 1, 0), mk(2, 97, 9, 9, 8, 1, 1, 1, 0), 1, 1)


@pytest.mark.parametrize("shape,kernel", [((64, 64, 56, 56, 256), "bodahip_k1_stream_f32"), ((52, 96, 55, 55, 96), "bodahip_k1_quad_f32")], ids=["res2_64to256", "nin_cccp1"])
def test_k1_stream_auto_choice_matches_tiled_kernel(be, shape, kernel):
    """At the sizes where the planner picks a streaming kernel by itself (ResNet-50 res2 at B=64: 64 -> 256 chans on 56x56; NiN cccp1 / cccp2: 96 -> 96 on 55x55 planes, the
    16-bytes-per-lane kernel) its output is bit-identical to the tiled kernel's (which the oracle pins at small sizes), and a spec never captures shapes it does not cover."""
    rtc = be.rtc
    op = _conv_op(*shape, 1, 1, 1, 0)
    anno = add_codegen_annotations(op, OpTune()); fn = anno.get_func_name()
    rtc.compile([RtcFuncInfo("k1s_auto", "", [a for a, _ in NATIVE_ARGS[fn]], anno)])
    am = {}
    for an, io in NATIVE_ARGS[fn]:
        if io == "REF": am[an] = RtcArg.ref(anno.get_dims(an)); continue
        rtc.create_var_with_dims("k1a_" + an, anno.get_dims(an)); am[an] = RtcArg.var("k1a_" + an)
        if io == "IN": rtc.run(gd.gen_call("Convolution", an, "k1a_" + an, anno.get_dims(an), 5, 0.0))
    try:
        rtc.run(RtcFuncCall("k1s_auto", am)); rtc.finish_and_sync()
        assert rtc.last_launch()["kernel"] == kernel
        a = rtc.copy_var_to_nda("k1a_out")
        rtc.set_tune("k1_stream", "off"); rtc.set_var_to_zero("k1a_out")
        rtc.run(RtcFuncCall("k1s_auto", am)); rtc.finish_and_sync()
        assert rtc.last_launch()["kernel"] == "bodahip_conv_f32"
        assert np.array_equal(a, rtc.copy_var_to_nda("k1a_out")) and float(np.abs(a).max()) > 0
    finally:
        rtc.set_tune("k1_stream", "")
        for an, io in NATIVE_ARGS[fn]:
            if io != "REF": rtc.release_var("k1a_" + an)
        rtc.release_func("k1s_auto"); rtc.release_per_call_id_data()


RDEC_CASES = [  # (B, C, H, W, OC, K, S): strided, unpadded, wide kernels -> the row-decimated LDS patch (gemm_conv_f32.hip -DRDEC=1): AlexNet / NiN conv1 form, tiles spanning
    (3, 3, 39, 39, 96, 11, 4),       # several images, one output row per image, ragged out_chans, odd / even K (row sets per step x KW), non-square planes, stride 2 and 3
    (20, 3, 227, 227, 96, 11, 4),
    (7, 5, 20, 33, 40, 7, 3),
    (2, 4, 16, 64, 33, 6, 2),
    (9, 1, 11, 11, 7, 11, 4),
    (5, 2, 30, 19, 130, 8, 2),
]


@pytest.mark.parametrize("case", RDEC_CASES, ids=lambda c: "x".join(str(v) for v in c))
def test_row_decimated_patch_bit_exact(be, case, monkeypatch):
    """Strided convolutions without padding (conv1 layers) through the row-decimated patch: same bits as the oracle and as the row-gather kernel it replaces."""
    B, C, H, W, OC, K, S = case
    op = _conv_op(B, C, H, W, OC, K, K, S, 0)
    monkeypatch.delenv("BODAHIP_RDEC", raising=False)
    outs, prc = _run(be, op, 5, include_ins=True)
    assert "_w" in prc.launch["cfg"] and prc.launch["kernel"] in ("bodahip_conv_f32", "bodahip_conv_big_f32")   # (round 6: at bench-like sizes with 96-multiples of out_chans the staging-wave kernel's row-decimated form)
    from boda_amd.rtc import explain_plan
    assert ("-DRDEC=1" in explain_plan(add_codegen_annotations(op, OpTune()))) == ((H - K) // S + 1 > 1)   # (a single output row stays on the row gather)
    want = bo.conv_fwd(outs["in"], outs["filts"], outs["biases"], (S, S), (0, 0), True)
    assert np.array_equal(want, outs["out"]), SsdsDiff.of(want, outs["out"]).basic_str()
    monkeypatch.setenv("BODAHIP_RDEC", "off")
    ref, _ = _run(be, op, 5)
    assert np.array_equal(ref["out"], outs["out"])
    for t in ("96x256x1x4x2", "64x256x1x4x2", "128x256x2x4x1"):
        monkeypatch.setenv("BODAHIP_RDEC", t)
        got, _ = _run(be, op, 5)
        assert np.array_equal(ref["out"], got["out"]), t


FC_CASES = [  # (B, C, H, W, OC, BODAHIP_FC = TMxTNxBKFxPF): ragged images / out_chans (one and several tiles each way), K tails (K % BKF != 0, fewer K tiles than the
    (5, 4, 3, 3, 7, "64x64x64x4"),      # ring and the stages hold), every tile shape, both K steps and ring depths, a 1x1 window (AlexNet fc7 form) and a real window (fc6 form)
    (70, 8, 2, 2, 130, "64x64x32x2"),
    (64, 256, 1, 1, 64, "64x64x64x2"),
    (3, 31, 4, 4, 65, "32x32x32x4"),
    (129, 12, 5, 5, 200, "64x32x64x2"),
    (20, 256, 6, 6, 96, "32x64x64x4"),
    (33, 64, 2, 2, 100, "32x32x64x2"),
]


@pytest.mark.parametrize("case", FC_CASES, ids=lambda c: "x".join(str(v) for v in c[:5]) + "_" + c[5])
@pytest.mark.parametrize("relu", [True, False])
def test_fc_kernel_bit_exact(be, case, relu, monkeypatch):
    """kernels/fc_f32.hip (whole-input windows: x-major LDS images, four k per operand read, staging between the MFMAs) forced onto small shapes: same bits as the oracle,
    as the tiled kernel, and a guard band around its channel slice of a wider output stays untouched."""
    rtc = be.rtc
    B, C, H, W, OC, spec = case
    op = _conv_op(B, C, H, W, OC, H, W, 1, 0)
    anno = add_codegen_annotations(op, OpTune())
    anno.nda_vals["conv_has_relu"].v = (int(relu),)
    fn = anno.get_func_name()
    rtc.compile([RtcFuncInfo("fc_conv", "", [a for a, _ in NATIVE_ARGS[fn]], anno)])
    x = bo.gen_conv_in(B, C, H, W); f = bo.gen_conv_filts(OC, C, H, W); b = bo.gen_conv_biases(OC)
    wide = Dims.make("float", img=B, chan=OC + 5, y=1, x=1)
    names = {"in": ("fc_in", anno.get_dims("in"), x), "filts": ("fc_f", anno.get_dims("filts"), f), "biases": ("fc_b", anno.get_dims("biases"), b),
             "out": ("fc_out", wide, np.full(wide.sizes, 7.0, np.float32))}
    for vn, d, arr in names.values():
        rtc.create_var_with_dims(vn, d); rtc.copy_nda_to_var(vn, arr)
    try:
        am = {an: RtcArg.var(names[an][0]) for an in names}
        am["stride"] = RtcArg.ref(anno.get_dims("stride")); am["in_pad"] = RtcArg.ref(anno.get_dims("in_pad"))
        am["out_chan_off"] = RtcArg.scalar(3, "uint32_t")
        monkeypatch.setenv("BODAHIP_FC", spec)
        rtc.run(RtcFuncCall("fc_conv", am)); rtc.finish_and_sync()
        assert rtc.last_launch()["kernel"] == "bodahip_fc_f32"
        got = rtc.copy_var_to_nda("fc_out")
        want = bo.conv_fwd(x, f, b, (1, 1), (0, 0), relu)
        assert np.array_equal(got[:, 3:3 + OC], want), SsdsDiff.of(want, got[:, 3:3 + OC]).basic_str()
        assert (got[:, :3] == 7).all() and (got[:, 3 + OC:] == 7).all()
        monkeypatch.setenv("BODAHIP_FC", "off")
        rtc.run(RtcFuncCall("fc_conv", am)); rtc.finish_and_sync()
        assert rtc.last_launch()["kernel"] == "bodahip_conv_f32" and np.array_equal(rtc.copy_var_to_nda("fc_out"), got)
    finally:
        for vn, _, _ in names.values():
            rtc.release_var(vn)
        rtc.release_func("fc_conv"); rtc.release_per_call_id_data()


def test_fc_kernel_is_the_planners_choice_for_alexnet_fc6_fc7(be, monkeypatch):
    """AlexNet fc6 / fc7 / fc8 at 256 images (256 tiles of 64 x 64; fc8: of 32 x 32) take the fc kernel by themselves; its output is bit-identical to the tiled kernel's."""
    rtc = be.rtc
    for (C, H, OC) in ((256, 6, 4096), (4096, 1, 4096), (4096, 1, 1000)):
        op = _conv_op(256, C, H, H, OC, H, H, 1, 0)
        anno = add_codegen_annotations(op, OpTune()); fn = anno.get_func_name()
        rtc.compile([RtcFuncInfo("fc_auto", "", [a for a, _ in NATIVE_ARGS[fn]], anno)])
        am = {}
        for an, io in NATIVE_ARGS[fn]:
            if io == "REF": am[an] = RtcArg.ref(anno.get_dims(an)); continue
            rtc.create_var_with_dims("fca_" + an, anno.get_dims(an)); am[an] = RtcArg.var("fca_" + an)
            if io == "IN": rtc.run(gd.gen_call("Convolution", an, "fca_" + an, anno.get_dims(an), 5, 0.0))
        try:
            monkeypatch.delenv("BODAHIP_FC", raising=False)
            rtc.run(RtcFuncCall("fc_auto", am)); rtc.finish_and_sync()
            assert rtc.last_launch()["kernel"] == "bodahip_fc_f32"
            a = rtc.copy_var_to_nda("fca_out")
            monkeypatch.setenv("BODAHIP_FC", "off"); rtc.set_var_to_zero("fca_out")
            rtc.run(RtcFuncCall("fc_auto", am)); rtc.finish_and_sync()
            assert rtc.last_launch()["kernel"] == "bodahip_conv_f32"
            assert np.array_equal(a, rtc.copy_var_to_nda("fca_out")) and float(np.abs(a).max()) > 0
        finally:
            for an, io in NATIVE_ARGS[fn]:
                if io != "REF": rtc.release_var("fca_" + an)
            rtc.release_func("fc_auto"); rtc.release_per_call_id_data()


WINO_CASES = [  # (B, C, H, W, OC, pad): odd and even planes, no / unit / double padding, 1-wide planes, ragged channel counts
    (3, 6, 10, 10, 12, 1), (2, 5, 13, 13, 7, 1), (5, 16, 7, 9, 33, 0), (1, 3, 3, 3, 4, 0), (2, 8, 4, 5, 8, 2), (9, 24, 14, 14, 40, 1), (2, 1, 6, 1, 2, 1),
    (3, 19, 9, 11, 68, 1), (4, 40, 13, 13, 128, 1),   # several 8-channel stages (ragged last one), several 64-out_chan blocks and 64-tile blocks of the fused kernel
]


@pytest.mark.parametrize("case", WINO_CASES, ids=lambda c: "x".join(str(v) for v in c))
@pytest.mark.parametrize("relu", [True, False])
def test_winograd_conv_within_reference_tolerance(be, case, relu):
    """Opt-in F(2x2,3x3) Winograd path (tune conv_algo=winograd): not bit-exact by construction; held to the tolerance the reference
    itself applies to Winograd results (3x3 cudnn_conv: mrd < 2e-3, src/rtc_prof.cc:317-319,436) -- measured ~1e-6 -- and it writes
    exactly its channel range of a wider output."""
    rtc = be.rtc
    B, C, H, W, OC, P = case
    op = _conv_op(B, C, H, W, OC, 3, 3, 1, P)
    anno = add_codegen_annotations(op, OpTune())
    anno.nda_vals["conv_has_relu"].v = (int(relu),)
    fn = anno.get_func_name()
    rtc.compile([RtcFuncInfo("wino_conv", "", [a for a, _ in NATIVE_ARGS[fn]], anno)])
    x = bo.gen_conv_in(B, C, H, W); f = bo.gen_conv_filts(OC, C, 3, 3); b = bo.gen_conv_biases(OC)
    OH, OW = H + 2 * P - 2, W + 2 * P - 2
    wide = Dims.make("float", img=B, chan=OC + 3, y=OH, x=OW)
    names = {"in": ("wn_in", anno.get_dims("in"), x), "filts": ("wn_f", anno.get_dims("filts"), f), "biases": ("wn_b", anno.get_dims("biases"), b),
             "out": ("wn_out", wide, np.full(wide.sizes, 7.0, np.float32))}
    for vn, d, arr in names.values():
        rtc.create_var_with_dims(vn, d); rtc.copy_nda_to_var(vn, arr)
    try:
        am = {an: RtcArg.var(names[an][0]) for an in names}
        am["stride"] = RtcArg.ref(anno.get_dims("stride")); am["in_pad"] = RtcArg.ref(anno.get_dims("in_pad"))
        am["out_chan_off"] = RtcArg.scalar(1, "uint32_t")
        rtc.set_tune("conv_algo", "winograd_all")
        rtc.run(RtcFuncCall("wino_conv", am)); rtc.finish_and_sync()
        assert rtc.last_launch()["kernel"] == "bodahip_conv_winograd_f32"
        got = rtc.copy_var_to_nda("wn_out")
        want = bo.conv_fwd(x, f, b, (1, 1), (P, P), relu)
        sd = SsdsDiff.of(want, got[:, 1:1 + OC])
        assert not sd.has_nan() and sd.mrd < 2e-3, sd.basic_str()
        assert sd.mrd < 5e-4, sd.basic_str()      # these small layers measure 6e-5 .. 1.4e-4 on the reference's U(-5,5) data (K = 9*in_chan <= 360 terms)
        assert (got[:, :1] == 7).all() and (got[:, 1 + OC:] == 7).all()
        if OC % 4 == 0:   # the opt-in fused kernel == the three-kernel pipeline bit for bit: same transform expressions, same ascending-in_chan MFMA chain per position
            os.environ["BODAHIP_WINO_FUSED"] = "1"
            try:
                rtc.copy_nda_to_var("wn_out", names["out"][2])
                rtc.run(RtcFuncCall("wino_conv", am)); rtc.finish_and_sync()
                assert rtc.last_launch()["kernel"] == "bodahip_conv_winograd_fused_f32"
                assert np.array_equal(got, rtc.copy_var_to_nda("wn_out"))
            finally:
                del os.environ["BODAHIP_WINO_FUSED"]
    finally:
        rtc.set_tune("conv_algo", "")
        for vn, _, _ in names.values():
            rtc.release_var(vn)
        rtc.release_func("wino_conv"); rtc.release_per_call_id_data()


def test_winograd_by_func_name_through_op_tune(be):
    """op_tune hip_algo=winograd -> func hip_conv_winograd: the Winograd path for that function only (3x3 / stride 1), the exact direct
    kernel for shapes it does not cover, nothing sticky on the backend."""
    rtc = be.rtc
    wino = OpTune.parse("(hip_algo=winograd)")
    assert add_codegen_annotations(_conv_op(2, 6, 12, 12, 10, 3, 3, 1, 1), wino).get_func_name() == "hip_conv_winograd"
    outs, _ = _run(be, _conv_op(2, 6, 12, 12, 10, 3, 3, 1, 1), tune=wino)
    assert rtc.last_launch()["kernel"] == "bodahip_conv_winograd_f32"
    want = bo.conv_fwd(bo.gen_conv_in(2, 6, 12, 12), bo.gen_conv_filts(10, 6, 3, 3), bo.gen_conv_biases(10), (1, 1), (1, 1), True)
    sd = SsdsDiff.of(want, outs["out"])
    assert not sd.has_nan() and 0 < sd.mrd < 5e-4, sd.basic_str()
    outs, _ = _run(be, _conv_op(2, 6, 12, 12, 10, 5, 5, 2, 1), tune=wino)
    assert rtc.last_launch()["kernel"] == "bodahip_conv_f32"
    assert np.array_equal(outs["out"], bo.conv_fwd(bo.gen_conv_in(2, 6, 12, 12), bo.gen_conv_filts(10, 6, 5, 5), bo.gen_conv_biases(10), (2, 2), (1, 1), True))
    outs, _ = _run(be, _conv_op(2, 6, 12, 12, 10, 3, 3, 1, 1))
    assert rtc.last_launch()["kernel"] == "bodahip_conv_f32" and np.array_equal(outs["out"], want)


def test_tile_travels_with_the_function(be):
    """op_tune hip_tile is carried by the annotated op (str_val hip_tile): the backend applies it to that function's calls only -- per-layer
    tuned tiles (tools/tune_tiles.py, the op-tuner idea of src/op-tuner.cc) need no backend-wide state -- and results stay bit-exact."""
    rtc = be.rtc
    op = _conv_op(4, 24, 14, 14, 48, 3, 3, 1, 1)
    x = bo.gen_conv_in(4, 24, 14, 14); f = bo.gen_conv_filts(48, 24, 3, 3); b = bo.gen_conv_biases(48)
    want = bo.conv_fwd(x, f, b, (1, 1), (1, 1), True)
    for vn, d, arr in (("tt_in", op.get_dims("in"), x), ("tt_f", op.get_dims("filts"), f), ("tt_b", op.get_dims("biases"), b), ("tt_out", op.get_dims("out"), None)):
        rtc.create_var_with_dims(vn, d)
        if arr is not None: rtc.copy_nda_to_var(vn, arr)
    try:
        cfgs = {}
        for name, tune in (("tt_tiled", OpTune(hip_tile="128x128x16x2x2x2")), ("tt_auto", OpTune())):
            anno = add_codegen_annotations(op, tune)
            assert ("hip_tile" in anno.str_vals) == (name == "tt_tiled")
            rtc.compile([RtcFuncInfo(name, "", [a for a, _ in NATIVE_ARGS[anno.get_func_name()]], anno)])
        am = {"in": RtcArg.var("tt_in"), "filts": RtcArg.var("tt_f"), "biases": RtcArg.var("tt_b"), "out": RtcArg.var("tt_out"),
              "stride": RtcArg.ref(op.get_dims("stride")), "in_pad": RtcArg.ref(op.get_dims("in_pad"))}
        for name in ("tt_tiled", "tt_auto", "tt_tiled"):
            rtc.set_var_to_zero("tt_out")
            rtc.run(RtcFuncCall(name, am)); rtc.finish_and_sync()
            cfgs.setdefault(name, []).append(rtc.last_launch()["cfg"])
            assert np.array_equal(rtc.copy_var_to_nda("tt_out"), want), name
        assert all(c.startswith("128x128x") and c.endswith("_w2x2") for c in cfgs["tt_tiled"]) and not cfgs["tt_auto"][0].startswith("128x128x")
    finally:
        for vn in ("tt_in", "tt_f", "tt_b", "tt_out"): rtc.release_var(vn)
        rtc.release_func("tt_tiled"); rtc.release_func("tt_auto"); rtc.release_per_call_id_data()


@pytest.mark.parametrize("net,batch", [("googlenet_conv", 64), ("resnet-50", 64), ("nin", 128)])
def test_config5_layers_at_bench_batch_prefix_invariance(be, net, batch):
    """Full-size property at BASELINE config 5's per-GPU batch (64): the planner picks other tiles / kernels there than at batch 2
    (32x256 and 32x128 four-wave patch tiles, the streaming 1x1 kernel, 64x64 with a 32-deep K step ...), so every distinct layer is run
    at B=64 and out[:2] must equal the oracle's B=2 result bit for bit (inputs are a hash of the flat index: the first two images of
    the B=64 input are the B=2 input); the last image must be finite and not all zero."""
    import bench
    ops_of = (lambda b: bench.nin_ops(b)) if net == "nin" else (lambda b: bench.net_conv_ops(net, b))   # (nin at 128: config 4's per-GPU batch)
    big = {}
    for op in ops_of(batch):
        big.setdefault(op.to_str(), op)
    small = {}
    for op in ops_of(2):
        small.setdefault(op.to_str(), op)
    assert len(big) == len(small) >= 9
    kernels = set()
    for ob, os_ in zip(big.values(), small.values()):
        outs, prc = _run(be, ob, 5)
        want = bo.run_op(os_, 5)["out"]
        assert np.array_equal(want, outs["out"][:2]), (ob.to_str(), prc.launch["kernel"], prc.launch["cfg"], SsdsDiff.of(want, outs["out"][:2]).basic_str())
        last = outs["out"][-1]
        assert np.isfinite(last).all() and last.max() > 0
        kernels.add(prc.launch["kernel"] + " " + prc.launch["cfg"].split("x")[0] + "x" + prc.launch["cfg"].split("x")[1])
    assert len(kernels) >= 2, kernels


# ---------------------------------------------------------------------------------------------------------------
# tolerance mode (op_tune hip_exact=0): deterministic K slices on tile-starved long-K layers, Winograd on 3x3 / stride-1 layers.  Both re-associate
# the fp32 sum, so they are held to the reference's bound for re-associating kernels (mrd < 2e-3, src/rtc_prof.cc:317-319,436) against the
# bit-exact chain -- NOT to its 2e-4 default (:161): on the reference's own U(-5,5) data a K = 9216 dot product carries ~1e-2 of absolute
# rounding error in ANY association (the single chain included), which is 8.6e-4 of the smaller outputs (measured, fc6) -- and, for the K
# slices, to the exact fp64 result: no farther from it than the reference's chain is.  At the benched batch.
# ---------------------------------------------------------------------------------------------------------------
def test_tolerance_mode_at_bench_batch(be):
    import bench
    big, small = bench.alexnet_b256_ops(256), bench.alexnet_b256_ops(2)
    seen = set()
    for i in (2, 5, 6, 7):    # conv3 (3x3: Winograd), fc6 / fc7 / fc8 (K = 9216 / 4096 / 4096 on 256 / 256 / 64 tiles: K slices)
        outs, prc = _run(be, big[i], 5, tune=OpTune(hip_exact=0))
        assert prc.op.str_vals.get("hip_exact") == "0"
        want = bo.run_op(small[i], 5)["out"]
        sd = SsdsDiff.of(want, outs["out"][:2])
        assert not sd.has_nan() and sd.mrd < MRD_REASSOC, (prc.launch, sd.basic_str())
        if i == 2:
            assert "winograd" in prc.launch["kernel"], prc.launch
        else:
            assert "_s" in prc.launch["cfg"], prc.launch
            g = small[i].conv_geom(); K = g["C"] * g["KH"] * g["KW"]
            x = bo.gen_conv_in(2, g["C"], g["H"], g["W"]).reshape(2, K).astype(np.float64)
            f = bo.gen_conv_filts(g["OC"], g["C"], g["KH"], g["KW"]).reshape(g["OC"], K).astype(np.float64)
            exact64 = np.maximum(x @ f.T + bo.gen_conv_biases(g["OC"]).astype(np.float64), 0.0).reshape(want.shape)
            err_slices = np.abs(outs["out"][:2].astype(np.float64) - exact64).max(); err_chain = np.abs(want.astype(np.float64) - exact64).max()
            assert err_slices <= 1.25 * err_chain + 1e-6, (i, err_slices, err_chain)
        seen.add(prc.launch["kernel"])
        exact, prc_e = _run(be, big[i], 5)                                  # the default stays the bit-exact plan
        assert "_s" not in prc_e.launch["cfg"] and "winograd" not in prc_e.launch["kernel"] and np.array_equal(want, exact["out"][:2])
    assert len(seen) == 2


@pytest.mark.parametrize("net", ["googlenet_conv", "resnet-50"])
def test_tolerance_mode_every_layer_of_config5_nets_at_bench_batch(be, net):
    """hip_exact=0 over every distinct GoogLeNet / ResNet-50 layer at 64 images: whatever the planner re-associates there (K slices from K = 512 on tile-starved
    layers, Winograd on 3x3 / stride-1 layers with >= 96 channels) stays inside the reference's 2e-3; layers it leaves alone stay bit-exact."""
    import bench
    big, small = {}, {}
    for op in bench.net_conv_ops(net, 64):
        big.setdefault(op.to_str(), op)
    for op in bench.net_conv_ops(net, 2):
        small.setdefault(op.to_str(), op)
    n_sliced = n_wino = 0
    for ob, os_ in zip(big.values(), small.values()):
        outs, prc = _run(be, ob, 5, tune=OpTune(hip_exact=0))
        want = bo.run_op(os_, 5)["out"]
        sd = SsdsDiff.of(want, outs["out"][:2])
        re_assoc = ("_s" in prc.launch["cfg"]) or ("winograd" in prc.launch["kernel"])
        n_sliced += "_s" in prc.launch["cfg"]; n_wino += "winograd" in prc.launch["kernel"]
        assert not sd.has_nan() and sd.mrd < MRD_REASSOC, (ob.to_str(), prc.launch, sd.basic_str())
        if not re_assoc:
            assert np.array_equal(want, outs["out"][:2]), (ob.to_str(), prc.launch)
    assert n_sliced >= 1 and n_wino >= 2, (n_sliced, n_wino)


# ---------------------------------------------------------------------------------------------------------------
# bf16 kernels at the BENCHED sizes (BASELINE config 5: 64 images per GPU; config 3's fc layers at 256; sgemm-ops-full >= 2048).
# Parity is unpinned for bf16 by construction (the reference has none): the stated bound is  mrd < 1e-3 * max(1, sqrt(K/2400))
# against the oracle fed the same bf16-rounded operands, enforced here where the bench runs.
# ---------------------------------------------------------------------------------------------------------------
def _bf16_bound(K):
    return MRD_BF16 * max(1.0, (K / 2400.0) ** 0.5)


def _bf16_prefix_check(be, big_op, small_op):
    """big_op through hip_conv_bf16 at the benched batch; out[:2] against the oracle's batch-2 result on bf16-rounded operands (the
    inputs are a hash of the flat index: the first two images of the big input ARE the batch-2 input; filters / biases are equal)."""
    outs, prc = _run(be, big_op, 5, tune=OpTune(hip_dtype="bf16"))
    assert prc.op.get_func_name() == "hip_conv_bf16"
    g = small_op.conv_geom()
    ins = bo.run_op(small_op, 5)
    want = bo.conv_fwd(bo.to_bf16(ins["in"]), bo.to_bf16(ins["filts"]), ins["biases"], (g["SY"], g["SX"]), (g["PY"], g["PX"]), True)
    got = outs["out"][:2]
    sd = SsdsDiff.of(want, got)
    K = g["C"] * g["KH"] * g["KW"]
    assert not sd.has_nan() and sd.mrd < _bf16_bound(K), (big_op.to_str(), prc.launch["kernel"], prc.launch["cfg"], K, sd.basic_str())
    assert _nrms(ins["out"], got) < 1e-2            # and informationally against the exact fp32 result
    last = outs["out"][-1]
    assert np.isfinite(last).all() and last.max() > 0
    return prc.launch, sd.mrd / _bf16_bound(K)


@pytest.mark.parametrize("net", ["googlenet_conv", "resnet-50"])
def test_config5_every_layer_bf16_at_bench_batch(be, net):
    """Every distinct GoogLeNet / ResNet-50 layer at B=64 through hip_conv_bf16.  At this size the bf16 planner takes paths the small
    edge shapes never reach: default split-K on tile-starved long-K layers, 128x128 patch16 tiles, the space-to-depth conv1 front end,
    the 1x1 / table gathers -- each must have been taken at least once over the two nets."""
    import bench
    big, small = {}, {}
    for op in bench.net_conv_ops(net, 64):
        big.setdefault(op.to_str(), op)
    for op in bench.net_conv_ops(net, 2):
        small.setdefault(op.to_str(), op)
    assert len(big) == len(small) >= 20
    seen, worst = set(), 0.0
    for ob, os_ in zip(big.values(), small.values()):
        launch, rel = _bf16_prefix_check(be, ob, os_)
        worst = max(worst, rel)
        seen.add(launch["kernel"] + ("+splitk" if "_s" in launch["cfg"] else ""))
    print(f"{net}: bf16 kernels taken at B=64: {sorted(seen)}; worst mrd / bound = {worst:.3f}")
    assert any(k.startswith("bodahip_conv_patch_bf16") and "s2d" not in k for k in seen), seen      # channel-innermost LDS patch (3x3 / 5x5)
    assert any("s2d" in k for k in seen), seen                                                       # conv1 through space-to-depth
    assert any(k.startswith("bodahip_conv_bf16") for k in seen), seen                                # 1x1 / table gather kernel
    if net == "googlenet_conv":
        assert any(k.endswith("+splitk") for k in seen), seen                                        # inception 5a/5b 1x1 on 7x7 maps, aux heads


@pytest.mark.parametrize("layer", ALEXNET_B256[5:])
def test_alexnet_fc_layers_bf16_at_b256(be, layer):
    """AlexNet fc6 / fc7 / fc8 at B=256 in bf16: the ipconv-shaped operands, split over K by default (K = 9216 / 4096)."""
    C, H, W, OC, K, S, P = layer
    launch, rel = _bf16_prefix_check(be, _conv_op(256, C, H, W, OC, K, K, S, P), _conv_op(2, C, H, W, OC, K, K, S, P))
    assert launch["kernel"].startswith("bodahip_conv_bf16")
    if C * K * K >= 4096:
        assert "_s" in launch["cfg"], launch


@pytest.mark.parametrize("n", [2048, 3072, 4096, 5120, 6144, 7168, 8192, 10240, 12288])
def test_sgemm_full_sizes_bf16_row_sample(be, n):
    """The sgemm-ops-full sizes >= 2048 through hip_sgemm_bf16 on the reference's mode-5 data (mode 600's exact answer 1000 m + n needs
    more than bf16's 8 bits from 256 up): 16 rows of c spread over the matrix (first / last rows of the first, a middle and the last
    tile row) against an fp64 contraction of the bf16-rounded operands.  Bound: 1e-3 * max(1, K/2400) -- linear in K here, not the
    sqrt(K) of the conv layers: on U(-5,5) data the partial sums grow like sqrt(K) * 8.3 (2500 and more at K = 12288, fp32 ulp 2.4e-4)
    and the K/16 roundings of the fp32 accumulator random-walk over that, so the absolute error on the near-zero outputs that set mrd
    grows like K.  Measured on MI355X: 2.5e-4 (3072) 3.2e-4 (4096) 4.5e-4 (5120) 7.0e-4 (6144) 8.8e-4 (7168) 7.1e-4 (8192) 8.3e-4 (10240)
    2.5e-3 (12288).  It is the fp32 accumulator, not the bf16 operands: the rounded operands are what the fp64 reference is fed."""
    outs, prc = _run(be, _sgemm_op(n, n, n), 5, tune=OpTune(hip_dtype="bf16"))
    assert prc.op.get_func_name() == "hip_sgemm_bf16"
    rows = sorted({0, 1, 31, 32, 127, 128, 255, 256, n // 2 - 1, n // 2, n // 2 + 37, n - 257, n - 256, n - 129, n - 2, n - 1})
    a = bo.to_bf16(bo.gen_sgemm_a(n, n, 5)[:, rows]).astype(np.float64); b = bo.to_bf16(bo.gen_sgemm_b(n, n, 5)).astype(np.float64)
    want = (a.T @ b).astype(np.float32)
    got = outs["c"][rows]
    sd = SsdsDiff.of(want, got)
    bound = MRD_BF16 * max(1.0, n / 2400.0)
    print(f"sgemm bf16 {n}^3 [{prc.launch['cfg']}]: mrd {sd.mrd:.3e} (bound {bound:.3e})")
    assert not sd.has_nan() and sd.mrd < bound, (n, prc.launch["cfg"], sd.basic_str())
    assert float(np.abs(outs["c"][-1]).max()) > 0 and np.isfinite(outs["c"]).all()


def test_outputs_of_2gib_or_more_are_unsupported(be):
    """The epilogue masks lanes past the last column with byte offset 2^31, which the buffer range check only drops while the output
    ends below it: outputs of 2 GiB or more are refused (unsup_err), not silently corrupted."""
    op = _sgemm_op(23200, 23200, 4)   # c = 2.15 GB
    with pytest.raises(UnsupErr):
        _run(be, op, 5)


def test_ipconv_shapes_through_the_lds_dma_kernel_bit_exact(monkeypatch):
    """The exact fp32 variant of the LDS-DMA kernel (kernels/conv_nhwc_bf16.hip, IN_F32: b128 fragment reads feeding two 32x32x2 MFMAs in ascending
    k) on the shapes whose operands are k-contiguous in the reference layout (output 1x1, kernel == whole input).  Opt-in (measured slower than the
    gather kernel, native_plan.cc: plan_ipconv_dma) -- forced here and held to bit-exact equality with the oracle like every fp32 kernel."""
    monkeypatch.setenv("BODAHIP_IPCONV_DMA", "force")   # (also for shapes with fewer tiles than the planner would send there)
    rtc = make_rtc("(be=hip)", 0); rtc.init()
    b = OpsBackend(rtc)
    try:
        for shape in [(256, 64, 6, 6, 512, 6, 6, 1, 0), (200, 1024, 1, 1, 1000, 1, 1, 1, 0), (130, 37, 4, 4, 300, 4, 4, 1, 0)]:
            op = _conv_op(*shape)
            outs, prc = _run(b, op, 5, include_ins=True)
            assert prc.launch["kernel"] == "bodahip_conv_nhwc_f32", prc.launch
            g = op.conv_geom()
            want = bo.conv_fwd(outs["in"], outs["filts"], outs["biases"], (1, 1), (0, 0), True)
            assert np.array_equal(want, outs["out"]), (shape, SsdsDiff.of(want, outs["out"]).basic_str())
    finally:
        rtc.finish_and_sync(); rtc.close()
