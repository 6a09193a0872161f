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
    assert s.get_dims("filts") == Dims(("in_grp", "y", "x", "out_chan", "in_chan8"), (2, 4, 4, 64, 8), "bfloat16")    # (stride 1 now: the input-patch kernel's filter form)
    assert add_codegen_annotations(op, OpTune(hip_dtype="bf16", hip_layout="nhwc", hip_patch=0)).get_dims("filts") == Dims(("out_chan", "y", "x", "in_chan"), (64, 4, 4, 16), "bfloat16")
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
    assert "-DBK=64" in plan((64, 512, 7, 7, 512, 3, 3, 1, 1), hip_patch=0)
    # more than one tap, stride 1 in x: the annotation asks for the F' filter form and the function binds the LDS input-patch kernel; filter fragments straight from
    # global memory (ADIRECT), wave tiles of 32 out_chans x 64 or 128 pels, tile by the MFMA work of the busiest SIMD
    pk = plan((64, 512, 7, 7, 512, 3, 3, 1, 1))
    assert pk[0] == "bodahip_conv_nhwc_patch_bf16" and pk[1].startswith("64x128x144_w2x2") and "-DCG=2" in pk and "-DCIN=512" in pk and "-DADIRECT=1" in pk   # (7-wide rows padded to a pitch of 23 chunks: 2 groups per step keep two workgroups per CU)
    assert plan((64, 64, 56, 56, 192, 3, 3, 1, 1))[1].startswith("64x256x144_w2x2")        # 2352 tiles: the widest pel tile (2 of the 8 groups per step: two workgroups per CU fit the LDS)
    assert plan((64, 256, 14, 14, 256, 3, 3, 1, 1))[1].startswith("128x128x288_w4x1")      # 196 tiles of four 32 x 128 wave tiles: one round
    big = plan((256, 96, 27, 27, 256, 5, 5, 1, 2))                                            # 1458 tiles of 64 x 128 wave tiles: half the operand bytes per MFMA,
    assert big[1].startswith("128x256x400_w2x2") and "-DPF=4" in big                          # four fragments in flight keep 128 accumulators at two waves per SIMD
    assert plan((64, 160, 7, 7, 320, 3, 3, 1, 1))[1].startswith("32x128x144_w1x4")         # 3136 pels: 250 small tiles rather than 125 on half the CUs
    assert "-DCG=2" in plan((64, 32, 28, 28, 96, 5, 5, 1, 2))                                # 5x5: two channel groups per K step (50 k-slots, no zero slot)
    assert "-DCG=1" in plan((64, 24, 14, 14, 64, 5, 5, 1, 2))                                # three groups: 3 x 26 slots rather than 2 x 50
    assert "-DCG=2" in plan((64, 112, 14, 14, 224, 3, 3, 1, 1))                              # 14 groups: 7 x 18 slots rather than 4 x 36
    assert plan((64, 256, 56, 56, 64, 1, 1, 1, 0))[0] == plan((64, 128, 28, 28, 128, 3, 3, 2, 1))[0] == "bodahip_conv_nhwc_bf16"   # 1x1, and stride 2 in x: implicit GEMM
    assert plan((64, 128, 4, 4, 1024, 4, 4, 1, 0))[0] == "bodahip_conv_nhwc_bf16"            # whole-input kernel (an fc layer): implicit GEMM + K slices
    fc = plan((64, 2048, 1, 1, 1000, 1, 1, 1, 0))                                # 64 output rows, K = 2048: K slices, reduced inside the launch (round 5: KSL)
    assert any(o.startswith("-DKSL=") for o in fc) and "-DSPLITK=1" not in fc and "_s" in fc[1]
    big = plan((64, 1024, 14, 14, 256, 1, 1, 1, 0))
    assert "-DSPLITK=1" not in big and not any(o.startswith("-DKSL=") for o in big)        # 3.2 M outputs: the fp32 partial tiles would cost more than they save
    # small maps (14 x 14 / 7 x 7 at 64 images) are latency-bound: 64-deep K steps, ragged taps included (480 channels = 7.5 steps); large maps keep the 32-deep rules
    assert "-DBK=64" in plan((64, 512, 14, 14, 128, 1, 1, 1, 0)) and "-DBK=64" in plan((64, 480, 14, 14, 96, 1, 1, 1, 0)) and "-DBK=32" in plan((64, 192, 28, 28, 96, 1, 1, 1, 0))
    assert "-DOUT_F32=1" in plan((2, 64, 8, 8, 64, 3, 3, 1, 1), hip_out="f32")
    t = plan((2, 64, 8, 8, 64, 3, 3, 1, 1), hip_tile="64x64x32x2x2x2x2x32x3", hip_patch=0)
    assert t[1].startswith("64x64x32_w2x2_s2") and "-DNBUF=3" in t
    for kw in (dict(hip_patch=0), {}):
        with pytest.raises(UnsupErr):
            plan((2, 64, 8, 8, 64, 3, 3, 1, 1), hip_tile="48x64x32x1x2", **kw)           # not a multiple of the MFMA tile
    with pytest.raises(UnsupErr):
        plan((2, 64, 8, 8, 64, 3, 3, 1, 1), hip_tile="32x64x32x1x4x2x1x32x3", hip_patch=0)  # uneven loads per wave: no deep ring


def test_sibling_group_annotation():
    """hip_conv_nhwc_grp: the members' filters / biases stacked along out_chan, each member padded to grp.pad rows; tiles no taller than the padding."""
    t = OpTune(hip_dtype="bf16", hip_layout="nhwc")
    mem = [add_codegen_annotations(_conv_op(64, 192, 28, 28, oc, 1, 1, 1, 0), t) for oc in (96, 16, 64)]
    ga = nhwc.annotate_group(mem)
    assert ga.get_func_name() == "hip_conv_nhwc_grp" and ga.get_dims("grp") == Dims(("m0", "m1", "m2", "pad"), (96, 16, 64, 32), "none")
    assert ga.get_dims("filts") == Dims(("out_chan", "y", "x", "in_chan"), (192, 1, 1, 192), "bfloat16") and ga.get_dims("biases").sizes == (192,)
    assert nhwc.group_row_offsets(ga.get_dims("grp")) == [0, 96, 128] and nhwc.group_arg_names(3)[-3:] == ["out_0", "out_1", "out_2"]
    assert [ga.get_dims(f"out_{m}").dsz("chan") for m in range(3)] == [96, 16, 64]
    assert nhwc.group_pad([192, 48, 384]) == 128 and nhwc.group_pad([128, 32, 128]) == 64 and nhwc.group_pad([16, 16]) == 32
    p = rtc.explain_plan(ga).split()
    assert p[0] == "bodahip_conv_nhwc_bf16" and "-DGROUPS=1" in p and p[1].startswith("32x") and "-DSPLITK=1" not in p
    assert rtc.parse_op_native(ga.to_str()) == ga.to_str() and rtc.prebuild(ga) > 0
    with pytest.raises(UnsupErr):
        nhwc.annotate_group(mem[:1])                                                                        # a group has 2..4 members
    with pytest.raises(UnsupErr):
        nhwc.annotate_group([mem[0], add_codegen_annotations(_conv_op(64, 192, 28, 28, 32, 1, 1, 2, 0), t)])   # another stride: not siblings
    with pytest.raises(UnsupErr):
        nhwc.annotate_group([add_codegen_annotations(_conv_op(64, 96, 28, 28, oc, 3, 3, 1, 1), t) for oc in (64, 64)])   # patch-form filters are not stacked


def test_kernel_and_layout_passes_compile_for_gfx950_without_a_device():
    assert rtc.compile_offline(nhwc.XPOSE_SRC, use_cache=True) > 0
    for shape, kw in [((2, 24, 28, 28, 64, 5, 5, 1, 2), {}), ((2, 3, 64, 64, 24, 11, 11, 4, 5), dict(hip_out="f32")), ((4, 2048, 1, 1, 1000, 1, 1, 1, 0), {})]:
        assert rtc.prebuild(add_codegen_annotations(_conv_op(*shape), OpTune(hip_dtype="bf16", hip_layout="nhwc", **kw))) > 0


def test_sibling_fixtures_agree_with_the_net_graphs():
    """boda_amd/data/nets/<net>-conv-bottoms.txt (bench.py --group-siblings): the convolutions that share a bottom blob.  GoogLeNet: exactly the groups ConvPipeFwd
    finds in the pipe (nine inception modules x {1x1, 3x3-reduce, 5x5-reduce}); ResNet-50: the four stage-start pairs branch1 + branch2a."""
    import os
    from boda_amd.conv_pipe import googlenet_conv
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    def groups(net):
        nb = [l.split() for l in open(os.path.join(root, "boda_amd", "data", "nets", f"{net}-conv-bottoms.txt")).read().splitlines() if l.strip()]
        by = {}
        for n, b in nb:
            by.setdefault(b, []).append(n)
        return nb, sorted(tuple(v) for v in by.values() if len(v) > 1)
    nb, gg = groups("googlenet_conv")
    cp = googlenet_conv(2)
    convs = [o for o in cp.ops if o.type == "Convolution"]
    assert [n for n, _ in nb] == [o.tag for o in convs] and [b for _, b in nb] == [o.bot for o in convs]
    assert len(gg) == 9 and all(len(g) == 3 for g in gg)
    nb, gr = groups("resnet-50")
    assert len(nb) == 54 and gr == sorted((f"res{s}a_branch1", f"res{s}a_branch2a") for s in (2, 3, 4, 5))


def test_wide_planes_are_annotated_for_the_kernel_that_can_run_them():
    """The layout of `filts` binds the kernel, so the annotation asks for the input-patch form only where plan_conv_nhwc_patch finds a tile whose patch fits
    the 160 KB of LDS (nhwc.patch_min_lds mirrors its bound); wider planes -- and space-to-depth layers on such planes -- stay on the implicit GEMM.  Every
    annotated op of the sweep must be plannable, and the Python bound must flip exactly where the planner's does."""
    flips = 0
    for (kh, pad) in ((3, 1), (5, 2), (2, 0), (7, 3)):
        last = None
        for w in list(range(13, 1200, 37)) + [224, 227, 500, 512, 850, 1024]:
            for out in ("", "f32"):
                op = _conv_op(1, 16, 8 + kh, w, 32, kh, kh, 1, pad)
                a = add_codegen_annotations(op, OpTune(hip_dtype="bf16", hip_layout="nhwc", hip_out=out))
                patch = a.get_dims("filts").has("in_grp")
                plan = rtc.explain_plan(a)                                     # raises UnsupErr if annotation and planner disagree
                assert plan.startswith("bodahip_conv_nhwc_patch_bf16 " if patch else "bodahip_conv_nhwc_bf16 "), (kh, w, plan[:60])
                # the bound is tight: forcing the patch form where the annotation declined it is refused by the planner
                if not patch:
                    forced = add_codegen_annotations(op, OpTune(hip_dtype="bf16", hip_layout="nhwc", hip_out=out))
                    from boda_amd.op import Nda
                    forced.nda_vals["filts"] = Nda(dims=nhwc.patch_filts_dims(op.get_dims("filts")), tn="bfloat16")
                    with pytest.raises(UnsupErr, match="no tile fits the LDS"):
                        rtc.explain_plan(forced)
                if out == "":
                    flips += int(last is not None and last != patch and w > 224); last = patch if w not in (224, 227, 500, 512, 850, 1024) else last
    assert flips >= 2                                                          # the sweep crosses the bound (5x5 near 500 columns, 7x7 earlier)
    # space-to-depth on a plane too wide for its patch: same form of `in`, implicit-GEMM filters
    wide = add_codegen_annotations(_conv_op(1, 3, 64, 4100, 32, 7, 7, 2, 3), OpTune(hip_dtype="bf16", hip_layout="nhwc"))
    assert wide.has("nhwc_s2d") and not wide.get_dims("filts").has("in_grp") and rtc.explain_plan(wide).startswith("bodahip_conv_nhwc_bf16 ")
    narrow = add_codegen_annotations(_conv_op(1, 3, 64, 224, 32, 7, 7, 2, 3), OpTune(hip_dtype="bf16", hip_layout="nhwc"))
    assert narrow.get_dims("filts").has("in_grp") and rtc.explain_plan(narrow).startswith("bodahip_conv_nhwc_patch_bf16 ")


def test_multi_problem_annotation_and_plan():
    """hip_conv_nhwc_multi: member args with the suffix _<m>, one tile shape for the launch (the kernel is specialised on the tile, not on any member's geometry), the C++
    op parser takes the (long) line, and the kernel cross-compiles."""
    shapes = [(64, 480, 14, 14, 192, 1, 1, 1, 0), (64, 832, 7, 7, 48, 1, 1, 1, 0), (64, 256, 56, 56, 128, 1, 1, 2, 0), (64, 24, 14, 14, 64, 5, 5, 1, 2)]
    annos = [add_codegen_annotations(_conv_op(*s), OpTune(hip_dtype="bf16", hip_layout="nhwc", hip_patch=0)) for s in shapes]
    m = nhwc.annotate_multi(annos)
    assert m.get_func_name() == "hip_conv_nhwc_multi" and m.get_dims("multi").dsz("n") == 4 and m.get_u32("conv_has_relu") == 1 and not m.has("relu_mask")
    assert nhwc.multi_arg_names(2) == ["multi", "filts_0", "biases_0", "in_0", "stride_0", "in_pad_0", "out_0", "filts_1", "biases_1", "in_1", "stride_1", "in_pad_1", "out_1"]
    assert m.get_dims("in_2") == annos[2].get_dims("in") and m.get_dims("stride_2").sizes == (2, 2) and m.get_dims("kern_sz_3").sizes == (5, 5)
    assert rtc.parse_op_native(m.to_str()) == m.to_str()
    plan = rtc.explain_plan(m)
    assert plan.startswith("bodahip_conv_nhwc_multi_bf16 64x128x64_w2x2_p2 ") and "-DMINW=3" in plan and "-DOUT_F32=0" in plan and "-DCIN" not in plan and "-DKH" not in plan      # nothing of a member's geometry
    assert rtc.explain_plan(m, tile="32x128x64x1x4x2x1x32x2").startswith("bodahip_conv_nhwc_multi_bf16 32x128x64_w1x4_p2 ")
    with pytest.raises(UnsupErr):
        rtc.explain_plan(m, tile="64x96x64x2x2")                                                                                             # (a wave tile is whole 32x32 MFMA blocks)
    assert rtc.prebuild(m) > 8000
    # members that differ in ReLU: a mask instead of the common flag
    annos[1].nda_vals["conv_has_relu"].v = (0,)
    mm = nhwc.annotate_multi(annos)
    assert mm.get_u32("conv_has_relu") == 0 and mm.get_u32("relu_mask") == 0b1101


def test_level_set_annotation_and_wrapper_kernel():
    """hip_conv_nhwc_set: members keep their own plans (input-patch or implicit GEMM by the dims of their filts); one wrapper kernel holds one instantiation of the
    kernel sources per distinct plan and cross-compiles; members with another workgroup size stay launches of their own."""
    T = OpTune(hip_dtype="bf16", hip_layout="nhwc")
    shapes = [(64, 96, 14, 14, 208, 3, 3, 1, 1), (64, 16, 14, 14, 48, 5, 5, 1, 2), (64, 480, 14, 14, 64, 1, 1, 1, 0)]      # icp3: 3x3, 5x5, pool projection
    annos = [add_codegen_annotations(_conv_op(*s), T) for s in shapes]
    assert all(nhwc.set_eligible(a) for a in annos) and [nhwc.multi_eligible(a) for a in annos] == [False, False, True]
    m = nhwc.annotate_set(annos)
    assert m.get_func_name() == "hip_conv_nhwc_set" and m.get_dims("multi").dsz("n") == 3
    assert m.get_dims("filts_0").names == ("in_grp", "y", "x", "out_chan", "in_chan8") and m.get_dims("filts_2").names == ("out_chan", "y", "x", "in_chan")
    assert rtc.parse_op_native(m.to_str()) == m.to_str()
    plan = rtc.explain_plan(m)
    assert plan.startswith("bodahip_conv_nhwc_set variants=3 ") and plan.count("bodahip_conv_nhwc_patch_bf16:") == 2 and plan.count("bodahip_conv_nhwc_bf16:") == 1
    # every member's plan is the one its own launch takes; the wrapper lists them longest tile first (the order the run builds it in: one code object for both)
    assert sorted(part.split(":")[1] for part in plan.split()[2:]) == sorted(rtc.explain_plan(a).split()[1] for a in annos)
    assert rtc.prebuild(m) > 20000
    # two members on the same plan share one instantiation
    twin = nhwc.annotate_set([annos[2], add_codegen_annotations(_conv_op(*shapes[2]), T)])
    assert rtc.explain_plan(twin).startswith("bodahip_conv_nhwc_set variants=1 ")
    with pytest.raises(UnsupErr):
        nhwc.annotate_set(annos[:1])
    s2d = add_codegen_annotations(_conv_op(64, 3, 224, 224, 64, 7, 7, 2, 3), T)
    with pytest.raises(UnsupErr):
        nhwc.annotate_set([annos[0], s2d])
