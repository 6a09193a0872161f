"""Host logic of the channels-last bf16 variant (no GPU): annotation (kernel dims + <arg>_ref dims, the reference's transposed-operand
protocol, src/cnn_op.cc:142-330 / src/rtc_prof.cc:92-121), the planner's choice, device-less compilation of the kernel and the layout passes."""
import pytest

from boda_amd import nhwc, rtc
from boda_amd.cnn_op import NATIVE_ARGS, OpTune, add_codegen_annotations
from boda_amd.op import Dims, UnsupErr, parse_op


def _conv_op(B, C, H, W, OC, KH, KW, S, P):
    OH = (H + 2 * P - KH) // S + 1; OW = (W + 2 * P - KW) // S + 1
    return parse_op(f"(str_vals=(type=Convolution),nda_vals=(biases=(dims=(out_chan={OC})),filts=(dims=(out_chan={OC},in_chan={C},y={KH},x={KW})),"
                    f"in=(dims=(img={B},chan={C},y={H},x={W})),in_pad=(tn=none,dims=(y={P},x={P})),kern_sz=(tn=none,dims=(y={KH},x={KW})),"
                    f"out=(dims=(img={B},chan={OC},y={OH},x={OW})),out_chans=(tn=uint32_t,v={OC}),stride=(tn=none,dims=(y={S},x={S}))))")


def test_annotation_gives_kernel_dims_and_keeps_reference_dims():
    op = _conv_op(64, 3, 224, 224, 64, 7, 7, 2, 3)
    a = add_codegen_annotations(op, OpTune(hip_dtype="bf16", hip_layout="nhwc", hip_s2d=0))
    assert a.get_func_name() == "hip_conv_nhwc" and "hip_conv_nhwc" in NATIVE_ARGS
    assert a.get_dims("in") == Dims(("img", "y", "x", "chan"), (64, 224, 224, 8), "bfloat16")          # 3 channels stored as 8 (zero pad)
    assert a.get_dims("filts") == Dims(("out_chan", "y", "x", "in_chan"), (64, 7, 7, 8), "bfloat16")
    assert a.get_dims("out") == Dims(("img", "y", "x", "chan"), (64, 112, 112, 64), "bfloat16")
    for an in ("in", "filts", "out"):
        assert a.get_dims(an + "_ref") == op.get_dims(an)
    assert a.get_dims("biases").tn == "float" and a.get_u32("conv_has_relu") == 1
    assert op.flops() == 2 * 64 * 112 * 112 * 64 * 3 * 49                   # credit is the op's own 2MNK, not the padded kernel's
    f = add_codegen_annotations(op, OpTune(hip_dtype="bf16", hip_layout="nhwc", hip_out="f32"))
    assert f.get_dims("out").tn == "float"
    # the C++ op-line parser of the backend takes the annotated line (new type name included) and gives the same canonical text back
    assert rtc.parse_op_native(a.to_str()) == a.to_str()
    # conv1-type layers (stride 2 on 3 channels) go space-to-depth by default: 2x2 pixel blocks -> 12 (stored: 16) channels, the 7x7 / 2 / pad 3
    # layer becomes a 4x4 / 1 / pad 0 one on a 115 x 115 map (pad 3 rounded up to 4: 224 + 4 + ... -> (112 + 4 - 1)); originals kept as <arg>_ref
    s = add_codegen_annotations(op, OpTune(hip_dtype="bf16", hip_layout="nhwc"))
    assert s.get_dims("in") == Dims(("img", "y", "x", "chan"), (64, 115, 115, 16), "bfloat16")
    assert s.get_dims("filts") == Dims(("out_chan", "y", "x", "in_chan"), (64, 4, 4, 16), "bfloat16")
    assert s.get_dims("stride").sizes == (1, 1) and s.get_dims("in_pad").sizes == (0, 0) and s.get_dims("kern_sz").sizes == (4, 4)
    assert s.get_dims("stride_ref").sizes == (2, 2) and s.get_dims("in_pad_ref").sizes == (3, 3) and s.get_dims("kern_sz_ref").sizes == (7, 7)
    assert (s.get_u32("nhwc_s2d"), s.get_u32("nhwc_s2d_pry"), s.get_u32("nhwc_s2d_prx")) == (2, 4, 4)
    s.conv_geom()   # (the annotated op is itself a consistent convolution)
    # a 3x3 / 2 layer on 64 channels is not a conv1-type layer
    assert not add_codegen_annotations(_conv_op(2, 64, 14, 14, 128, 3, 3, 2, 1), OpTune(hip_dtype="bf16", hip_layout="nhwc")).has("nhwc_s2d")
    # without the layout knob nothing changes: fp32 NCHW tensors, operands rounded while staging
    assert add_codegen_annotations(op, OpTune(hip_dtype="bf16")).get_func_name() == "hip_conv_bf16"


def test_planner_choices():
    plan = lambda shape, **kw: rtc.explain_plan(add_codegen_annotations(_conv_op(*shape), OpTune(hip_dtype="bf16", hip_layout="nhwc", **kw)), tile=kw.get("hip_tile", "")).split()
    p = plan((64, 256, 56, 56, 64, 1, 1, 1, 0))
    assert p[0] == "bodahip_conv_nhwc_bf16" and "-DCIN=256" in p and "-DOUT_F32=0" in p
    assert "-DBK=32" in plan((64, 64, 56, 56, 256, 1, 1, 1, 0))                 # short K: 32-deep steps, deeper ring
    assert "-DBK=64" in plan((64, 512, 7, 7, 512, 3, 3, 1, 1))
    fc = plan((64, 2048, 1, 1, 1000, 1, 1, 1, 0))                                # 64 output rows, K = 2048: K slices + reduce pass
    assert "-DSPLITK=1" in fc and "_s" in fc[1]
    assert "-DSPLITK=1" not in plan((64, 1024, 14, 14, 256, 1, 1, 1, 0))        # 3.2 M outputs: the fp32 partial tiles would cost more than they save
    assert "-DOUT_F32=1" in plan((2, 64, 8, 8, 64, 3, 3, 1, 1), hip_out="f32")
    t = plan((2, 64, 8, 8, 64, 3, 3, 1, 1), hip_tile="64x64x32x2x2x2x2x32x3")
    assert t[1].startswith("64x64x32_w2x2_s2") and "-DNBUF=3" in t
    with pytest.raises(UnsupErr):
        plan((2, 64, 8, 8, 64, 3, 3, 1, 1), hip_tile="48x64x32x1x2")           # not a multiple of the MFMA tile
    with pytest.raises(UnsupErr):
        plan((2, 64, 8, 8, 64, 3, 3, 1, 1), hip_tile="32x64x32x1x4x2x1x32x3")  # uneven loads per wave: no deep ring


def test_kernel_and_layout_passes_compile_for_gfx950_without_a_device():
    assert rtc.compile_offline(nhwc.XPOSE_SRC, use_cache=True) > 0
    for shape, kw in [((2, 24, 28, 28, 64, 5, 5, 1, 2), {}), ((2, 3, 64, 64, 24, 11, 11, 4, 5), dict(hip_out="f32")), ((4, 2048, 1, 1, 1000, 1, 1, 1, 0), {})]:
        assert rtc.prebuild(add_codegen_annotations(_conv_op(*shape), OpTune(hip_dtype="bf16", hip_layout="nhwc", **kw))) > 0
