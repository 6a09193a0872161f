"""Parity tests (-m gpu) of the channels-last bf16 convolution (func hip_conv_nhwc, kernels/conv_nhwc_bf16.hip), through the C ABI and the
reference's `<arg>_ref` + xpose protocol (boda_amd/ops_prof.py, src/rtc_prof.cc:92-121): data is generated in the reference layout,
layout passes fill the kernel's bf16 tensors, the result is transposed back and compared with the CPU oracle fed the same bf16-rounded
operands.  The reference has no bf16: parity is UNPINNED by construction; the stated bounds are
    float output:     mrd < 1e-3 * max(1, sqrt(K / 2400))                       (the per-layer bound of every bf16 kernel here; empirical)
    bfloat16 output:  the float result rounded once more: |got - want| <= 2^-8 * |want| + the float bound
  and, since round 5, the bound DERIVED from the arithmetic, checked element by element beside them (see _derived_limit):
    |got - want| <= 2 (K + 1) 2^-24 * sum_k |in_k| |filts_k|  (+ 2^-8 |want| for a bf16 output)
  measured against it per layer of both config-5 nets at the bench batch: profiles/r05_bf16_error_table.txt (tools/bf16_error_table.py).
Nothing here reads /root/reference."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from boda_amd.cnn_op import OpTune, add_codegen_annotations
from boda_amd.digest import SsdsDiff
from boda_amd.op import parse_op
from boda_amd.ops_prof import OpsBackend, profile_rcg_call
from boda_amd.rtc import make_rtc
from oracle import boda_oracle as bo

NHWC_F32 = dict(hip_dtype="bf16", hip_layout="nhwc", hip_out="f32")
NHWC = dict(hip_dtype="bf16", hip_layout="nhwc")


@pytest.fixture(scope="module")
def be():
    rtc = make_rtc("(be=hip)", 0)
    rtc.init()
    b = OpsBackend(rtc)
    yield b
    rtc.finish_and_sync()
    rtc.close()


def _conv_op(B, C, H, W, OC, KH, KW, S, P):
    OH = (H + 2 * P - KH) // S + 1; OW = (W + 2 * P - KW) // S + 1
    return parse_op(f"(str_vals=(type=Convolution),nda_vals=(biases=(dims=(out_chan={OC})),filts=(dims=(out_chan={OC},in_chan={C},y={KH},x={KW})),"
                    f"in=(dims=(img={B},chan={C},y={H},x={W})),in_pad=(tn=none,dims=(y={P},x={P})),kern_sz=(tn=none,dims=(y={KH},x={KW})),"
                    f"out=(dims=(img={B},chan={OC},y={OH},x={OW})),out_chans=(tn=uint32_t,v={OC}),stride=(tn=none,dims=(y={S},x={S}))))")


def _bound(K):
    return 1e-3 * max(1.0, (K / 2400.0) ** 0.5)


def _run(be, op, tune):
    anno = add_codegen_annotations(op, tune)
    assert anno.get_func_name() == "hip_conv_nhwc"
    return profile_rcg_call(be, anno, 5, 0.0, 1, include_ins=True, tile=tune.hip_tile)


def _want(op, outs):
    g = op.conv_geom()
    return bo.conv_fwd(bo.to_bf16(outs["in"]), bo.to_bf16(outs["filts"]), outs["biases"], (g["SY"], g["SX"]), (g["PY"], g["PX"]), True), g["C"] * g["KH"] * g["KW"]


def _derived_limit(op, outs):
    """The bound that follows from the arithmetic alone (round 5; SURVEY section 8d asked for a K-derived one).  Kernel and oracle multiply the SAME bf16 operands -- the
    products are exact in fp32 -- and differ only in the order of K + 1 fp32 additions; either order is within (K + 1) u S of the exact sum, u = 2^-24,
    S = sum_k |in_k| |filts_k| + |bias| per output (the standard forward bound of recursive summation in any order; ReLU is 1-Lipschitz).  So, element by element,
        |got - want| <= 2 (K + 1) 2^-24 S          and for a bf16 output one more rounding:  + 2^-8 |want|.
    S comes from the oracle run on the absolute values.  K slices, MFMA-internal order, the patch kernel's tap-major order: all covered -- no fitted constant."""
    g = op.conv_geom(); K = g["C"] * g["KH"] * g["KW"]
    S = bo.conv_fwd(np.abs(bo.to_bf16(outs["in"])), np.abs(bo.to_bf16(outs["filts"])), np.abs(outs["biases"]), (g["SY"], g["SX"]), (g["PY"], g["PX"]), False).astype(np.float64)
    return 2.0 * (K + 1) * 2.0 ** -24 * S


def _check_f32(op, outs, prc):
    want, K = _want(op, outs)
    sd = SsdsDiff.of(want, outs["out"])
    assert not sd.has_nan() and sd.mrd < _bound(K), (op.to_str(), prc.launch["cfg"], K, sd.basic_str())
    err = np.abs(outs["out"].astype(np.float64) - want.astype(np.float64)); lim = _derived_limit(op, outs)
    assert (err <= lim).all(), (op.to_str(), prc.launch["cfg"], K, "derived bound", float((err / np.maximum(lim, 1e-300)).max()))
    return sd.mrd / _bound(K)


def _check_bf16(op, outs, prc):
    want, K = _want(op, outs)
    got = outs["out"]
    assert np.array_equal(bo.to_bf16(got), got)          # every value is a bf16
    err = np.abs(got.astype(np.float64) - want.astype(np.float64))
    lim = 2.0 ** -8 * np.abs(want.astype(np.float64)) + _bound(K) * np.maximum(1.0, np.abs(want.astype(np.float64)))
    assert np.isfinite(got).all() and (err <= lim).all(), (op.to_str(), prc.launch["cfg"], float((err / lim).max()))
    d = _derived_limit(op, outs)
    lim2 = 2.0 ** -8 * (np.abs(want.astype(np.float64)) + d) + d      # the derived form of the same statement (round-to-nearest-even: half a bf16 ulp <= 2^-8 of the fp32 result)
    assert (err <= lim2).all(), (op.to_str(), prc.launch["cfg"], "derived bound", float((err / np.maximum(lim2, 1e-300)).max()))


EDGE = [  # B, C, H, W, OC, KH, KW, S, P : 1x1 (stride 1 / 2, padded), 3x3, 5x5 / 2, 7x7 / 2, 11x11 / 4, whole-input kernels, ragged channels and maps
    (2, 19, 11, 11, 40, 1, 1, 1, 0), (3, 64, 14, 14, 128, 1, 1, 2, 0), (2, 8, 7, 7, 16, 1, 1, 1, 1), (1, 3, 12, 12, 16, 3, 3, 1, 1),
    (2, 5, 17, 13, 7, 5, 5, 2, 2), (3, 4, 9, 9, 33, 1, 1, 1, 0), (1, 8, 6, 6, 40, 6, 6, 1, 0), (2, 3, 35, 35, 96, 11, 11, 4, 0),
    (1, 16, 14, 14, 130, 7, 7, 2, 3), (5, 32, 7, 7, 64, 1, 1, 2, 0), (2, 6, 10, 10, 12, 3, 3, 1, 0), (1, 1, 5, 5, 1, 5, 5, 1, 2),
    (4, 20, 8, 8, 100, 3, 3, 1, 1), (2, 3, 40, 40, 16, 7, 7, 2, 3), (3, 4, 33, 31, 20, 5, 5, 1, 2), (2, 3, 64, 64, 24, 11, 11, 4, 5),
    (4, 96, 27, 27, 256, 5, 5, 1, 2), (3, 256, 13, 13, 384, 3, 3, 1, 1), (8, 256, 6, 6, 512, 6, 6, 1, 0), (3, 528, 14, 14, 160, 1, 1, 1, 0),
    (2, 112, 14, 14, 224, 3, 3, 1, 1), (2, 24, 28, 28, 64, 5, 5, 1, 2), (2, 832, 7, 7, 1000, 1, 1, 1, 0),
    (9, 96, 6, 3, 208, 7, 7, 1, 2), (3, 64, 1, 40, 64, 3, 3, 1, 1), (16, 32, 2, 2, 48, 3, 3, 1, 1),   # maps one or two positions wide / high (a fuzz find: the patch of a 4 x 1 map needs one workgroup per CU)
    (1, 8, 6, 520, 16, 5, 5, 1, 2), (1, 3, 16, 2100, 16, 7, 7, 2, 3)]   # planes too wide for the LDS patch of even one channel group: annotated for the implicit GEMM (plain and space-to-depth)


@pytest.mark.parametrize("shape", EDGE, ids=lambda s: "x".join(str(v) for v in s))
def test_nhwc_conv_float_out_vs_oracle(be, shape):
    op = _conv_op(*shape)
    outs, prc = _run(be, op, OpTune(**NHWC_F32))
    from boda_amd import nhwc
    g = op.conv_geom(); patch = nhwc.patch_eligible(g, True) or nhwc.s2d_geom(g) is not None   # more than one tap, stride 1 in x (also: the stride-1 space-to-depth form of a conv1-type layer): the LDS input-patch kernel
    if shape[3] >= 500:
        assert not nhwc.patch_eligible(g, True); patch = False                                    # (too wide for the patch: see nhwc.patch_min_lds)
    assert prc.launch["kernel"] == ("bodahip_conv_nhwc_patch_bf16" if patch else "bodahip_conv_nhwc_bf16"), prc.launch
    _check_f32(op, outs, prc)
    if patch:                                          # ... and the implicit-GEMM kernel on the same layer (op_tune hip_patch=0)
        outs, prc = _run(be, op, OpTune(hip_patch=0, **NHWC_F32))
        assert prc.launch["kernel"] == "bodahip_conv_nhwc_bf16"
        _check_f32(op, outs, prc)


@pytest.mark.parametrize("shape", EDGE[::2], ids=lambda s: "x".join(str(v) for v in s))
def test_nhwc_conv_bf16_out_vs_oracle(be, shape):
    op = _conv_op(*shape)
    outs, prc = _run(be, op, OpTune(**NHWC))
    _check_bf16(op, outs, prc)


@pytest.mark.parametrize("tile", ["128x128x64x2x2", "128x128x32x2x2", "64x128x64x1x4", "64x64x32x2x2", "32x128x32x1x4", "32x64x64x1x2", "64x256x32x2x4x1", "256x128x32x4x2x1",
                                  "128x128x64x2x2x2x1x32x3", "64x128x32x1x4x2x1x32x4", "64x64x64x2x2x2x3", "128x128x32x2x2x2x2x32x3", "32x64x32x1x2x2x5"])
def test_nhwc_conv_tiles_agree_with_oracle(be, tile):
    from boda_amd import nhwc
    for shape in [(3, 40, 15, 15, 100, 3, 3, 1, 1), (2, 64, 9, 9, 200, 1, 1, 1, 0), (2, 3, 33, 33, 48, 7, 7, 2, 3)]:
        op = _conv_op(*shape)
        outs, prc = _run(be, op, OpTune(hip_tile=tile, hip_patch=0, **NHWC_F32))
        assert prc.launch["cfg"].startswith(tile.split("x")[0] + "x" + tile.split("x")[1] + "x" + tile.split("x")[2]), prc.launch
        if len(tile.split("x")) >= 7 and int(tile.split("x")[6]) > 1:     # K slices, reduced inside the launch (round 5): the count is lowered until no slice is empty
            g = op.conv_geom(); bk = int(tile.split("x")[2]); nk = -(-(g["KH"] * g["KW"] * (-(-g["C"] // 8))) // (bk // 8)); want = min(int(tile.split("x")[6]), nk)
            while want > 1 and -(-nk // -(-nk // want)) != want:
                want -= 1
            import re
            got = int((re.search(r"_s(\d+)", prc.launch["cfg"]) or [0, "1"])[1])
            if nhwc.s2d_geom(g) is None:
                assert got == want, prc.launch
            else:       # (the space-to-depth form of a conv1-type layer has its own K steps: fewer taps on more channels)
                assert 1 <= got <= int(tile.split("x")[6]), prc.launch
        _check_f32(op, outs, prc)
        outs, prc = _run(be, op, OpTune(hip_tile=tile, hip_patch=0, **NHWC))
        _check_bf16(op, outs, prc)


@pytest.mark.parametrize("direct", [1, 0], ids=["direct", "staged"])
@pytest.mark.parametrize("tile", ["64x256x0x1x4", "64x128x0x1x4", "128x128x0x2x2", "32x128x0x1x4", "64x64x0x2x2", "32x64x0x1x2", "128x256x0x2x4x1", "64x128x0x2x2x1",
                                  "128x128x0x4x1", "64x256x0x2x2", "128x64x0x4x1", "256x128x0x4x1x1"])
def test_nhwc_patch_kernel_tiles_agree_with_oracle(be, tile, direct, monkeypatch):
    """kernels/conv_nhwc_patch_bf16.hip under forced tiles: 3x3 / 5x5 / 2x2 / 7x1-ish windows, padding 0..3, ragged channel counts (K tail of the channel groups),
    tiles that straddle image boundaries (maps of 15x15, 9x9, 5x5 pels against 64..256-pel tiles), ragged out_chans; float and bf16 outputs.  Both operand paths:
    filter fragments straight from global memory (the default; an odd count of k-slots per step, K steps past the end, out_chans past the end all read as zero
    through the buffer's range check) and both operands staged through the LDS."""
    monkeypatch.setenv("BODAHIP_NHWC_ADIRECT", str(direct))
    if not direct and tile not in ("64x256x0x1x4", "128x128x0x2x2", "32x64x0x1x2", "64x128x0x2x2x1"):
        pytest.skip("the staged form keeps four tiles under test")
    for shape in [(3, 40, 15, 15, 100, 3, 3, 1, 1), (5, 24, 9, 9, 70, 5, 5, 1, 2), (2, 8, 12, 12, 33, 2, 2, 1, 0), (7, 72, 5, 5, 64, 3, 3, 1, 1), (2, 16, 20, 11, 48, 7, 7, 1, 3), (40, 32, 3, 3, 64, 3, 3, 1, 1)]:
        op = _conv_op(*shape)
        outs, prc = _run(be, op, OpTune(hip_tile=tile, **NHWC_F32))
        assert prc.launch["kernel"] == "bodahip_conv_nhwc_patch_bf16" and prc.launch["cfg"].startswith(tile.split("x")[0] + "x" + tile.split("x")[1] + "x"), prc.launch
        _check_f32(op, outs, prc)
        outs, prc = _run(be, op, OpTune(hip_tile=tile, **NHWC))
        _check_bf16(op, outs, prc)


@pytest.mark.parametrize("net", ["googlenet_conv", "resnet-50"])
def test_config5_every_layer_nhwc_at_bench_batch(be, net):
    """BASELINE config 5 as benched (bench.py --workload googlenet|resnet50 --dtype bf16): every distinct layer at 64 images per GPU through
    hip_conv_nhwc; out[:2] against the oracle's batch-2 result on bf16-rounded operands (inputs are a hash of the flat index, so the first
    two images of the big input ARE the batch-2 input)."""
    import bench
    big, small = {}, {}
    for op in bench.net_conv_ops(net, 64):
        big.setdefault(op.to_str(), op)
    for op in bench.net_conv_ops(net, 2):
        small.setdefault(op.to_str(), op)
    assert len(big) == len(small) >= 20
    cfgs, worst = set(), 0.0
    for ob, os_ in zip(big.values(), small.values()):
        anno = add_codegen_annotations(ob, OpTune(**NHWC_F32))
        outs, prc = profile_rcg_call(be, anno, 5, 0.0, 1)
        ins = bo.run_op(os_, 5)
        g = os_.conv_geom(); K = g["C"] * g["KH"] * g["KW"]
        want = bo.conv_fwd(bo.to_bf16(ins["in"]), bo.to_bf16(ins["filts"]), ins["biases"], (g["SY"], g["SX"]), (g["PY"], g["PX"]), True)
        sd = SsdsDiff.of(want, outs["out"][:2])
        assert not sd.has_nan() and sd.mrd < _bound(K), (ob.to_str(), prc.launch["cfg"], K, sd.basic_str())
        worst = max(worst, sd.mrd / _bound(K))
        last = outs["out"][-1]
        assert np.isfinite(last).all() and last.max() > 0
        cfgs.add(prc.launch["cfg"])
        # ... and with bf16 OUTPUTS, which is what bench.py --dtype bf16 --layout nhwc times (the LDS-transposed 16-byte-store epilogue at the benched
        # tiles): the float result rounded once more -- |got - want| <= 2^-8 |want| + the float bound
        outs_b, prc_b = profile_rcg_call(be, add_codegen_annotations(ob, OpTune(**NHWC)), 5, 0.0, 1)
        got = outs_b["out"][:2]; w64 = want.astype(np.float64)
        # (same plan for both output types, except on the stems: the rolling-rows kernel writes bf16 only, float outputs stay on the input-patch kernel)
        assert (prc_b.launch["cfg"] == prc.launch["cfg"] or prc_b.launch["kernel"] == "bodahip_conv_nhwc_rows_bf16") and np.array_equal(bo.to_bf16(got), got), (prc_b.launch, prc.launch)
        err = np.abs(got.astype(np.float64) - w64); lim = 2.0 ** -8 * np.abs(w64) + _bound(K) * np.maximum(1.0, np.abs(w64))
        assert np.isfinite(outs_b["out"]).all() and (err <= lim).all(), (ob.to_str(), prc_b.launch["cfg"], float((err / lim).max()))
    print(f"{net}: hip_conv_nhwc tiles taken at B=64: {sorted(cfgs)}; worst mrd / bound = {worst:.3f}")
    assert len(cfgs) >= 2


@pytest.mark.parametrize("case", [dict(B=5, C=64, H=15, S=2, ocs=(96, 24, 130, 8), out="bf16"), dict(B=3, C=40, H=9, S=1, ocs=(64, 64), out="f32"),
                                   dict(B=2, C=256, H=14, S=2, ocs=(512, 128), out="bf16"), dict(B=4, C=24, H=7, S=1, ocs=(16, 200, 33), out="bf16", k=3)])
def test_fused_sibling_convs_equal_the_separate_calls(be, case):
    """hip_conv_nhwc_grp against its members run one by one on the same device tensors: bit-identical outputs (stride 1 / 2, ragged member sizes -- padded to 32 /
    64 / 128 rows --, float and bf16 outputs, a member writing a channel slice of a wider tensor; a 3x3 group through the implicit-GEMM form)."""
    from boda_amd import gen_data as gd, nhwc
    from boda_amd.rtc import RtcArg, RtcFuncCall, RtcFuncInfo
    rtc = be.rtc
    nhwc.ensure_compiled(rtc)
    B, C, H, S, k = case["B"], case["C"], case["H"], case["S"], case.get("k", 1)
    tune = OpTune(hip_patch=0, **(NHWC_F32 if case["out"] == "f32" else NHWC))
    annos = [add_codegen_annotations(_conv_op(B, C, H, H, oc, k, k, S, k // 2), tune) for oc in case["ocs"]]
    ga = nhwc.annotate_group(annos)
    made, funcs = [], []
    def mk(vn, dims):
        rtc.create_var_with_dims(vn, dims); made.append(vn)
    try:
        a0 = annos[0]
        mk("g_in_ref", a0.get_dims("in_ref")); mk("g_in", a0.get_dims("in"))
        rtc.run(gd.gen_call("Convolution", "in", "g_in_ref", a0.get_dims("in_ref"), 5, 0.0)); rtc.run(nhwc.xpose_call("in", "g_in_ref", "g_in", a0.get_dims("in_ref"), a0.get_dims("in"), a0))
        wide = a0.get_dims("out").sizes[:3] + (a0.get_dims("out").dsz("chan") + 24,)      # member 0 also writes channels [16, 16 + oc) of a wider tensor
        sep, F, Bv = [], np.zeros(ga.get_dims("filts").sizes, np.uint16), np.zeros(ga.get_dims("biases").sizes, np.float32)
        for m, (an, off) in enumerate(zip(annos, nhwc.group_row_offsets(ga.get_dims("grp")))):
            for arg in ("filts", "biases"):
                mk(f"g{m}_{arg}", an.get_dims(arg))
            mk(f"g{m}_filts_ref", an.get_dims("filts_ref"))
            rtc.run(gd.gen_call("Convolution", "filts", f"g{m}_filts_ref", an.get_dims("filts_ref"), 5, float(m))); rtc.run(gd.gen_call("Convolution", "biases", f"g{m}_biases", an.get_dims("biases"), 5, float(m)))
            rtc.run(nhwc.xpose_call("filts", f"g{m}_filts_ref", f"g{m}_filts", an.get_dims("filts_ref"), an.get_dims("filts"), an))
            odims = an.get_dims("out") if m else type(an.get_dims("out"))(an.get_dims("out").names, wide, an.get_dims("out").tn)
            mk(f"g{m}_out_sep", odims); mk(f"g{m}_out_grp", odims)
            rtc.finish_and_sync()
            f = rtc.copy_var_to_nda(f"g{m}_filts"); b = rtc.copy_var_to_nda(f"g{m}_biases")
            F[off:off + f.shape[0]] = f; Bv[off:off + b.shape[0]] = b
            fn = f"sep{m}"; rtc.compile([RtcFuncInfo(fn, "", ["filts", "biases", "in", "stride", "in_pad", "out"], an)]); funcs.append(fn)
            am = {"filts": RtcArg.var(f"g{m}_filts"), "biases": RtcArg.var(f"g{m}_biases"), "in": RtcArg.var("g_in"), "stride": RtcArg.ref(an.get_dims("stride")),
                  "in_pad": RtcArg.ref(an.get_dims("in_pad")), "out": RtcArg.var(f"g{m}_out_sep")}
            if m == 0:
                am["out_chan_off"] = RtcArg.scalar(16, "uint32_t")
            sep.append(RtcFuncCall(fn, am))
        mk("g_filts", ga.get_dims("filts")); mk("g_biases", ga.get_dims("biases"))
        rtc.copy_nda_to_var("g_filts", F); rtc.copy_nda_to_var("g_biases", Bv)
        rtc.compile([RtcFuncInfo("grp", "", nhwc.group_arg_names(len(annos)), ga)]); funcs.append("grp")
        gam = {"filts": RtcArg.var("g_filts"), "biases": RtcArg.var("g_biases"), "in": RtcArg.var("g_in"), "stride": RtcArg.ref(ga.get_dims("stride")),
               "in_pad": RtcArg.ref(ga.get_dims("in_pad")), "grp": RtcArg.ref(ga.get_dims("grp")), "out_chan_off_0": RtcArg.scalar(16, "uint32_t")}
        for m in range(len(annos)):
            gam[f"out_{m}"] = RtcArg.var(f"g{m}_out_grp")
        for c in sep:
            rtc.run(c)
        rtc.run(RtcFuncCall("grp", gam)); rtc.finish_and_sync()
        assert rtc.last_launch()["kernel"].startswith("bodahip_conv_nhwc_bf16(x")
        for m in range(len(annos)):
            a, b = rtc.copy_var_to_nda(f"g{m}_out_sep"), rtc.copy_var_to_nda(f"g{m}_out_grp")
            assert np.array_equal(a, b) and np.abs(a.astype(np.float64)).max() > 0, m
        w0 = rtc.copy_var_to_nda("g0_out_grp")
        assert not w0[..., :16].any() and not w0[..., 16 + case["ocs"][0]:].any()          # the guard channels of the wider tensor are untouched
    finally:
        rtc.finish_and_sync()
        for fn in funcs:
            rtc.release_func(fn)
        for vn in made:
            rtc.release_var(vn)
        rtc.release_per_call_id_data()


class _OpList:
    """A per-layer op list on the device, as bench.py builds it: every op its own function and its own tensors, inputs generated in the reference layout and brought
    to the kernel's layout by the layout passes.  member i: function name funcs[i], annotated op annos[i], call calls[i], vars `<pfx><i>_<arg>`."""

    def __init__(self, rtc, shapes, tune_of, pfx):
        from boda_amd import gen_data as gd, nhwc
        from boda_amd.cnn_op import NATIVE_ARGS
        from boda_amd.rtc import RtcArg, RtcFuncCall, RtcFuncInfo
        self.rtc, self.pfx, self.made, self.funcs, self.calls, self.annos, self.ops = rtc, pfx, [], [], [], [], []
        nhwc.ensure_compiled(rtc)
        for i, shp in enumerate(shapes):
            op = _conv_op(*shp); anno = add_codegen_annotations(op, tune_of(i)); fn = anno.get_func_name(); gen_fn = f"{pfx}_{i}"
            rtc.compile([RtcFuncInfo(gen_fn, "", [x for x, _ in NATIVE_ARGS[fn]], anno)]); self.funcs.append(gen_fn)
            am = {}
            for an, io in NATIVE_ARGS[fn]:
                if io == "REF":
                    am[an] = RtcArg.ref(anno.get_dims(an)); continue
                vn = f"{pfx}{i}_{an}"; rtc.create_var_with_dims(vn, anno.get_dims(an)); self.made.append(vn); am[an] = RtcArg.var(vn)
                if io == "IN":
                    rd = anno.get_dims(an + "_ref") if anno.has(an + "_ref") else anno.get_dims(an)
                    if rd != anno.get_dims(an):
                        rtc.create_var_with_dims(vn + "_ref", rd); self.made.append(vn + "_ref")
                        rtc.run(gd.gen_call("Convolution", an, vn + "_ref", rd, 5, 0.0)); rtc.run(nhwc.xpose_call(an, vn + "_ref", vn, rd, anno.get_dims(an), anno))
                    else:
                        rtc.run(gd.gen_call("Convolution", an, vn, rd, 5, 0.0))
            self.calls.append(RtcFuncCall(gen_fn, am)); self.annos.append(anno); self.ops.append(op)

    def out(self, i):
        return self.rtc.copy_var_to_nda(f"{self.pfx}{i}_out")

    def zero_outs(self):
        for i in range(len(self.calls)):
            self.rtc.set_var_to_zero(f"{self.pfx}{i}_out")

    def release(self):
        self.rtc.finish_and_sync()
        for vn in self.made:
            self.rtc.release_var(vn)
        for f in self.funcs:
            self.rtc.release_func(f)
        self.rtc.release_per_call_id_data()


def test_edge_free_graph_of_an_op_list_equals_call_by_call(be):
    """bench.py --graph --independent: a per-layer op list's ops share no tensor, so the captured graph gets no edges and its launches may overlap.  Same kernels, same
    arguments: every op's output must be bit-identical to its call-by-call run -- including three K-sliced members (partial tiles in the backend's ONE scratch buffer:
    multi-kernel calls stay chained inside and ordered among themselves) and an input-patch member."""
    rtc = be.rtc
    shapes = [(64, 832, 7, 7, 384, 1, 1, 1, 0), (64, 832, 7, 7, 128, 1, 1, 1, 0), (16, 192, 14, 14, 96, 1, 1, 1, 0), (8, 96, 14, 14, 208, 3, 3, 1, 1), (4, 64, 28, 28, 32, 1, 1, 2, 0),
              (64, 1024, 7, 7, 512, 1, 1, 1, 0)]
    # (members 0, 1 and 5: K slices forced through the function's own tile, so the scratch-sharing case is there whatever the planner thinks of the shape)
    ol = _OpList(rtc, shapes, lambda i: OpTune(hip_tile=("64x64x64x2x2x2x4" if i in (0, 1, 5) else ""), **NHWC), "efg")
    try:
        want, sliced = [], 0
        for i, c in enumerate(ol.calls):
            rtc.run(c); sliced += "_s" in rtc.last_launch()["cfg"]
            want.append(ol.out(i))
        assert sliced == 3
        rtc.finish_and_sync(); rtc.release_per_call_id_data()
        rtc.graph_begin()
        for c in ol.calls:
            rtc.run(c)
        gid = rtc.graph_end_deps([[] for _ in ol.calls])
        for rep in range(3):
            ol.zero_outs()
            rtc.graph_launch(gid); rtc.finish_and_sync()
            for i in range(len(ol.calls)):
                assert np.array_equal(ol.out(i), want[i]), (rep, shapes[i])
        rtc.graph_destroy(gid)
    finally:
        ol.release()


MULTI_SHAPES = [  # B, C, H, W, OC, KH, KW, S, P -- what a config-5 list holds besides its big layers, plus edges: ragged out_chans / pels, stride 2, padding, windows
    (8, 192, 28, 28, 96, 1, 1, 1, 0), (8, 192, 28, 28, 16, 1, 1, 1, 0), (16, 480, 14, 14, 192, 1, 1, 1, 0), (16, 528, 14, 14, 160, 1, 1, 1, 0), (64, 832, 7, 7, 384, 1, 1, 1, 0),
    (64, 832, 7, 7, 48, 1, 1, 1, 0), (4, 256, 56, 56, 128, 1, 1, 2, 0), (2, 1024, 14, 14, 2048, 1, 1, 2, 0), (3, 40, 15, 15, 100, 3, 3, 1, 1), (2, 24, 9, 13, 33, 5, 5, 1, 2),
    (2, 3, 33, 33, 48, 7, 7, 2, 3), (5, 19, 11, 11, 40, 1, 1, 1, 0), (64, 1024, 1, 1, 1000, 1, 1, 1, 0), (1, 8, 6, 6, 40, 6, 6, 1, 0), (2, 8, 7, 7, 16, 1, 1, 1, 1),
    (7, 128, 4, 4, 1024, 4, 4, 1, 0), (2, 112, 14, 14, 224, 3, 3, 1, 1), (1, 16, 3, 3, 8, 3, 3, 2, 1)]


@pytest.mark.parametrize("out,tile", [("", ""), ("f32", ""), ("", "64x128x64x2x2x2"), ("", "32x128x64x1x4x2x1x32x2"), ("f32", "128x64x32x2x2x2x1x32x4"), ("", "64x64x32x2x2x2x1x32x3")])
def test_multi_problem_launch_is_bit_identical_to_separate_launches(be, out, tile):
    """hip_conv_nhwc_multi (kernels/conv_nhwc_multi_bf16.hip): 18 independent convolutions of unlike geometry in ONE launch.  Per output the MFMA chain is that of the
    member's own hip_conv_nhwc launch on the implicit-GEMM kernel (ascending 16-k groups, zero beyond K): results must be bit-identical to the separate launches (run
    here without K slices and without the input-patch / space-to-depth forms, which sum in another order) -- and inside the stated bf16 bound of the oracle."""
    from boda_amd import nhwc
    from boda_amd.rtc import RtcArg, RtcFuncCall, RtcFuncInfo
    rtc = be.rtc
    tune = OpTune(hip_dtype="bf16", hip_layout="nhwc", hip_out=out, hip_patch=0, hip_s2d=0, hip_tile="64x64x64x2x2x2x1")
    ol = _OpList(rtc, MULTI_SHAPES, lambda i: tune, "mp")
    fn = "mp_multi"
    try:
        want = []
        for i, c in enumerate(ol.calls):
            rtc.run(c); ll = rtc.last_launch()
            assert ll["kernel"] == "bodahip_conv_nhwc_bf16" and "_s" not in ll["cfg"], ll
            want.append(ol.out(i))
        manno = nhwc.annotate_multi(ol.annos)
        if tile:
            manno.str_vals["hip_tile"] = tile
        else:
            manno.str_vals.pop("hip_tile", None)
        rtc.compile([RtcFuncInfo(fn, "", nhwc.multi_arg_names(len(ol.calls)), manno)])
        am = {"multi": RtcArg.ref(manno.get_dims("multi"))}
        for m, c in enumerate(ol.calls):
            for an in ("filts", "biases", "in", "stride", "in_pad", "out"):
                am[f"{an}_{m}"] = c.arg_map[an]
        ol.zero_outs()
        cid = rtc.run(RtcFuncCall(fn, am)); ll = rtc.last_launch()
        assert ll["kernel"] == f"bodahip_conv_nhwc_multi_bf16(x{len(ol.calls)})" and ll["flops"] > 0, ll
        if tile:
            assert ll["cfg"].startswith("x".join(tile.split("x")[:3])), ll
        rtc.finish_and_sync(); assert rtc.get_dur(cid, cid) > 0
        for i in range(len(ol.calls)):
            got = ol.out(i)
            assert np.array_equal(got, want[i]), (MULTI_SHAPES[i], ll["cfg"], int((got != want[i]).sum()))
        # ... and the same call inside a captured graph (descriptor table already on the device), replayed twice
        rtc.release_per_call_id_data()
        rtc.graph_begin(); rtc.run(RtcFuncCall(fn, am)); gid, n1 = rtc.graph_end(); assert n1 == 1
        for _ in range(2):
            ol.zero_outs(); rtc.graph_launch(gid); rtc.finish_and_sync()
            assert all(np.array_equal(ol.out(i), want[i]) for i in range(len(ol.calls)))
        rtc.graph_destroy(gid)
        # against the oracle on bf16-rounded operands (float outputs; a few members of each kind)
        if out == "f32":
            for i in (0, 6, 8, 10, 13, 17):
                op = ol.ops[i]; g = op.conv_geom()
                ins = bo.run_op(op, 5)
                ref = bo.conv_fwd(bo.to_bf16(ins["in"]), bo.to_bf16(ins["filts"]), ins["biases"], (g["SY"], g["SX"]), (g["PY"], g["PX"]), True)
                back = np.ascontiguousarray(np.transpose(want[i], (0, 3, 1, 2)))
                sd = SsdsDiff.of(ref, back)
                assert not sd.has_nan() and sd.mrd < _bound(g["C"] * g["KH"] * g["KW"]), (MULTI_SHAPES[i], sd.basic_str())
    finally:
        try:
            rtc.release_func(fn)
        except Exception:
            pass
        ol.release()


@pytest.mark.parametrize("out", ["", "f32"])
def test_level_set_is_bit_identical_to_separate_launches(be, out):
    """hip_conv_nhwc_set: an inception module's independent convolutions -- 3x3 and 5x5 on the input-patch kernel, 1x1 on the implicit GEMM -- in ONE launch, every
    member on ITS OWN specialised kernel code (the kernel sources instantiated per member inside a wrapper kernel built at run time).  Same code, same arguments,
    same tiles: the results are those of the members' own launches bit for bit; a member with another workgroup size is launched on its own by the same call."""
    from boda_amd import nhwc
    from boda_amd.rtc import RtcArg, RtcFuncCall, RtcFuncInfo
    rtc = be.rtc
    shapes = [(64, 96, 14, 14, 208, 3, 3, 1, 1), (64, 16, 14, 14, 48, 5, 5, 1, 2), (64, 480, 14, 14, 64, 1, 1, 1, 0), (8, 128, 28, 28, 192, 3, 3, 1, 1), (8, 256, 28, 28, 64, 1, 1, 1, 0),
              (3, 40, 15, 15, 100, 3, 3, 1, 1), (64, 160, 7, 7, 320, 3, 3, 1, 1), (64, 832, 7, 7, 128, 1, 1, 1, 0)]
    tune = OpTune(hip_dtype="bf16", hip_layout="nhwc", hip_out=out)
    ol = _OpList(rtc, shapes, lambda i: tune, "ls")
    fn = "ls_set"
    try:
        want, kernels = [], set()
        for i, c in enumerate(ol.calls):
            rtc.run(c); ll = rtc.last_launch(); kernels.add(ll["kernel"])
            assert "_s" not in ll["cfg"], (shapes[i], ll)          # (shapes the planner does not slice: the set never slices)
            want.append(ol.out(i))
        assert kernels == {"bodahip_conv_nhwc_bf16", "bodahip_conv_nhwc_patch_bf16"}
        sanno = nhwc.annotate_set(ol.annos)
        rtc.compile([RtcFuncInfo(fn, "", nhwc.multi_arg_names(len(ol.calls)), sanno)])
        am = {"multi": RtcArg.ref(sanno.get_dims("multi"))}
        for m, c in enumerate(ol.calls):
            for an in ("filts", "biases", "in", "stride", "in_pad", "out"):
                am[f"{an}_{m}"] = c.arg_map[an]
        ol.zero_outs()
        cid = rtc.run(RtcFuncCall(fn, am)); ll = rtc.last_launch()
        assert ll["kernel"].startswith("bodahip_conv_nhwc_set(x"), ll
        rtc.finish_and_sync(); assert rtc.get_dur(cid, cid) > 0
        for i in range(len(ol.calls)):
            got = ol.out(i)
            assert np.array_equal(got, want[i]), (shapes[i], ll, int((got != want[i]).sum()))
        rtc.release_per_call_id_data()
        rtc.graph_begin(); rtc.run(RtcFuncCall(fn, am)); gid, n1 = rtc.graph_end(); assert n1 == 1
        for _ in range(2):
            ol.zero_outs(); rtc.graph_launch(gid); rtc.finish_and_sync()
            assert all(np.array_equal(ol.out(i), want[i]) for i in range(len(ol.calls)))
        rtc.graph_destroy(gid)
    finally:
        try:
            rtc.release_func(fn)
        except Exception:
            pass
        ol.release()


def test_multi_problem_launch_refuses_what_it_cannot_run(be):
    from boda_amd import nhwc
    from boda_amd.op import UnsupErr
    patch = add_codegen_annotations(_conv_op(2, 64, 14, 14, 64, 3, 3, 1, 1), OpTune(**NHWC))
    plain = add_codegen_annotations(_conv_op(2, 64, 14, 14, 64, 1, 1, 1, 0), OpTune(**NHWC))
    f32 = add_codegen_annotations(_conv_op(2, 64, 14, 14, 64, 1, 1, 1, 0), OpTune(**NHWC_F32))
    assert nhwc.multi_eligible(plain) and not nhwc.multi_eligible(patch)
    with pytest.raises(UnsupErr):
        nhwc.annotate_multi([plain, patch])          # the input-patch form binds another kernel (another summation order)
    with pytest.raises(UnsupErr):
        nhwc.annotate_multi([plain, f32])            # one output type per launch
    with pytest.raises(UnsupErr):
        nhwc.annotate_multi([])


@pytest.mark.parametrize("shape,tile", [
    ((4, 832, 7, 7, 128, 1, 1, 1, 0), "64x64x64x2x2x2x4x32x3"), ((4, 832, 7, 7, 130, 1, 1, 1, 0), "32x128x64x1x4x2x8x32x3"), ((2, 1024, 1, 1, 1000, 1, 1, 1, 0), "64x64x64x2x2x2x16x32x3"),
    ((5, 160, 7, 7, 320, 3, 3, 1, 1), "64x64x0x2x2x2x2"), ((3, 192, 7, 7, 100, 3, 3, 1, 1), "128x64x0x4x1x2x3"), ((2, 48, 7, 7, 128, 5, 5, 1, 2), "64x128x0x2x2x2x2"),
    ((2, 128, 4, 4, 1024, 4, 4, 1, 0), "64x64x64x2x2x2x8x32x3")], ids=lambda v: v if isinstance(v, str) else "x".join(str(x) for x in v))
def test_k_slices_reduced_inside_the_launch(be, shape, tile):
    """Round 5 (KSL of kernels/conv_nhwc_bf16.hip / conv_nhwc_patch_bf16.hip): the grid is tiles x slices, every slice publishes its raw fp32 accumulators write-through,
    the workgroup that draws a tile's last ticket sums the slabs IN SLICE ORDER and runs the ordinary epilogue -- one kernel, no reduce pass.  Checked: against the oracle
    (float and bf16 outputs), run-to-run bitwise (the order of the sum does not depend on who arrives last), and launch after launch on the same workspace (the tickets
    are back at zero: eight launches in a row, the last one compared).  The slices of a tile run on different XCDs -- behind different, mutually incoherent L2s."""
    from boda_amd import nhwc
    op = _conv_op(*shape)
    first = None
    for rep in range(3):
        for tune in (OpTune(hip_tile=tile, **NHWC_F32), OpTune(hip_tile=tile, **NHWC)):
            anno = add_codegen_annotations(op, tune)
            outs, prc = profile_rcg_call(be, anno, 5, 0.0, 8 if rep == 2 else 1, include_ins=True, tile=tile)
            assert "_s" in prc.launch["cfg"] and prc.launch["kernel"] in ("bodahip_conv_nhwc_bf16", "bodahip_conv_nhwc_patch_bf16"), prc.launch
            (_check_f32 if tune.hip_out == "f32" else _check_bf16)(op, outs, prc)
            key = tune.hip_out
            if first is None or key not in first:
                first = dict(first or {}); first[key] = outs["out"].copy()
            else:
                assert np.array_equal(first[key], outs["out"]), (rep, key)


ROWS = [  # B, C, H, W, OC, KH, KW, S, P -- what the rolling-rows kernel takes: stride 1 after space-to-depth, more than one tap, at most 64 out_chans, a short K
    (3, 3, 61, 57, 64, 7, 7, 2, 3), (2, 3, 224, 224, 64, 7, 7, 2, 3), (5, 16, 19, 23, 40, 3, 3, 1, 1), (2, 8, 30, 20, 24, 3, 3, 1, 0), (7, 24, 9, 300, 64, 1, 3, 1, 1),
    (4, 8, 12, 12, 37, 5, 5, 1, 2), (300, 8, 6, 6, 16, 3, 3, 1, 1), (1, 32, 40, 9, 8, 2, 2, 1, 0), (3, 3, 35, 35, 33, 6, 6, 2, 2)]


@pytest.mark.parametrize("chunks", ["", "1", "3"], ids=["chunks-auto", "chunk-1", "chunks-3"])
@pytest.mark.parametrize("shape", ROWS, ids=lambda s: "x".join(str(v) for v in s))
def test_rolling_rows_kernel_vs_oracle_and_patch_kernel(be, shape, chunks, monkeypatch):
    """Round 5 (kernels/conv_nhwc_rows_bf16.hip): a workgroup walks down a run of output rows of one image with the filters in registers; the rows leave through an LDS
    ring.  Against the oracle (both bounds of this file); and against the patch kernel run on the same operands: the k-slot order is the same, so with an even tap count
    -- the space-to-depth stems -- the MFMA chains are the same whatever either planner picks, and the outputs are equal bit for bit.  Row runs of every length (one
    workgroup per image, three, as many as there are rows; a last run that is shorter), out_chans that do not fill the 64-row tile, padded and unpadded planes, a plane
    wider than a tile row's 256 positions is refused."""
    monkeypatch.setenv("BODAHIP_NHWC_ROWS", "1")
    if chunks:
        monkeypatch.setenv("BODAHIP_NHWC_ROWS_CHUNKS", chunks)
    op = _conv_op(*shape)
    outs, prc = _run(be, op, OpTune(**NHWC))
    if shape[3] == 300:      # 300 output positions per row: not this kernel's
        assert prc.launch["kernel"] != "bodahip_conv_nhwc_rows_bf16", prc.launch
        return
    assert prc.launch["kernel"] == "bodahip_conv_nhwc_rows_bf16", prc.launch
    _check_bf16(op, outs, prc)
    monkeypatch.setenv("BODAHIP_NHWC_ROWS", "0")
    outs2, prc2 = _run(be, op, OpTune(**NHWC))
    assert prc2.launch["kernel"] == "bodahip_conv_nhwc_patch_bf16", prc2.launch
    a = add_codegen_annotations(op, OpTune(**NHWC)); taps = a.get_dims("filts").dsz("y") * a.get_dims("filts").dsz("x")
    if taps % 2 == 0:
        assert np.array_equal(outs["out"], outs2["out"]), shape
    else:     # (an odd tap count pairs a step's last tap with the next group's first one or with a zero slot, by the channel groups per step: same sums, another order)
        assert float(np.max(np.abs(outs["out"].astype(np.float64) - outs2["out"]) / np.maximum(1.0, np.abs(outs2["out"])))) < 2.0 ** -7
