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
    from boda_amd.op import data_path
    for fn in sorted(glob.glob(os.path.join(golden_dir, "ops", "*.txt")) + glob.glob(data_path("ops", "*-conv-ops-b1.txt"))):
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


def _conv(B, C, H, OC, K, S=1, P=0, func="hip_conv"):
    from boda_amd.op import parse_op
    OH = (H + 2 * P - K) // S + 1
    return parse_op(f"(str_vals=(type=Convolution,func_name={func}),nda_vals=(biases=(dims=(out_chan={OC})),filts=(dims=(out_chan={OC},in_chan={C},y={K},x={K})),"
                    f"in=(dims=(img={B},chan={C},y={H},x={H})),in_pad=(tn=none,dims=(y={P},x={P})),kern_sz=(tn=none,dims=(y={K},x={K})),"
                    f"out=(dims=(img={B},chan={OC},y={OH},x={OH})),out_chans=(tn=uint32_t,v={OC}),stride=(tn=none,dims=(y={S},x={S})),conv_has_relu=(tn=uint32_t,v=1)))")


def test_planner_picks_the_documented_kernel_and_operand_mode_per_shape():
    """Variant / blocking selection is host logic (the counterpart of add_codegen_annotations' choice, src/cnn_op.cc:16-378): checked
    without a device through bodahip_explain_plan.  One shape per operand mode of DESIGN.md section 3."""
    ex = R.explain_plan
    def mode(plan, key): return re.search(rf"-D{key}=(\d+)", plan).group(1)
    p = ex(_conv(256, 96, 27, 256, 5, 1, 2))                      # AlexNet conv2: LDS input patch -- round 6: of the staging-wave kernel (two workgroups per CU, filters k-major from the scratch)
    assert p.startswith("bodahip_conv_big_f32 64x512x50_w1x8_p2_big") and mode(p, "J_MODE") == "7" and "-DCH=27" in p and "-DRELU=1" in p and "-DMINW=2" in p and "-DI_VW=0" in p
    assert "pels<163840+rest:64x192x50_w2x2_p2_big" in p                                                          # ... two-level: five whole rounds of 64 x 512 tiles (1280), the last 22 784 pels on 64 x 192
    os.environ["BODAHIP_CBIG_SPLIT"] = "off"
    try: assert "pels<" not in ex(_conv(256, 96, 27, 256, 5, 1, 2))
    finally: del os.environ["BODAHIP_CBIG_SPLIT"]
    assert ex(_conv(256, 256, 13, 384, 3, 1, 1)).startswith("bodahip_conv_big_f32 128x256x18_w2x4_p2_big") and "pels<" not in ex(_conv(256, 256, 13, 384, 3, 1, 1))   # AlexNet conv3: 507 tiles on 256 CUs: one launch
    q5 = ex(_conv(256, 384, 13, 256, 3, 1, 1)); assert q5.startswith("bodahip_conv_big_f32 128x256x18_w2x4_p2_big") and "pels<32768+rest:64x192x18_w2x2_p2_big" in q5   # conv5: one round of 128 x 256 + a tail (in one launch: 676 tiles of 64 x 256 = 2.64 per CU)
    os.environ["BODAHIP_CBIG"] = "off"
    try:
        q = ex(_conv(256, 96, 27, 256, 5, 1, 2)); assert q.startswith("bodahip_conv_f32 ") and mode(q, "J_MODE") == "7" and "-DCH=27" in q   # the round-3 kernel's patch form
    finally: del os.environ["BODAHIP_CBIG"]
    assert ex(_conv(256, 384, 6, 1024, 3, 1, 1)).startswith("bodahip_conv_big_f32 64x192x18_w2x2_p2_big")   # NiN conv4 (6 x 6 maps): four multiplying waves, 768 tiles = three per CU
    assert ex(_conv(128, 384, 6, 1024, 3, 1, 1)).startswith("bodahip_conv_big_f32 32x128x18_w1x4_p2_big")   # ... at 128 images: one-block wave tiles in place of the tiled kernel's 64 x 64 (1152 tiles = 4.5 per CU)
    assert ex(_conv(256, 1024, 6, 1024, 1)).startswith("bodahip_conv_big_f32 64x192x16_w2x2_p2_big") and "-DJ_MODE=5" in ex(_conv(256, 1024, 6, 1024, 1))   # cccp7: 1 x 1, K >= 384
    assert ex(_conv(64, 512, 28, 1024, 1, 2, 0)).startswith("bodahip_conv_f32 ")       # strided 1 x 1 (ResNet-50 res4a_branch1): stays on the tiled kernel
    assert ex(_conv(64, 128, 28, 192, 3, 1, 1)).startswith("bodahip_conv_f32 32x256")  # GoogLeNet 3x3 at 28 x 28: the tiled kernel's 32 x 256 tile beats the small staging-wave tiles
    assert ex(_conv(2, 24, 15, 100, 3, 1, 1)).startswith("bodahip_conv_f32 ")          # ... and so do small problems
    p = ex(_conv(256, 96, 27, 256, 5, 1, 2), tile="128x256x32x2x4x1x1x32x2x2")        # the kernel by its own tile string (tenth field 2): BKS = whole channels >= the tile's
    assert p.startswith("bodahip_conv_big_f32 128x256x50_w2x4_p2_big") and "-DMINW=1" in p
    p = ex(_conv(256, 384, 13, 384, 1), tile="128x128x16x2x4x2x1x32x2x2")              # 1 x 1: its plain form
    assert p.startswith("bodahip_conv_big_f32 128x128x16_w2x4_p2_big") and mode(p, "J_MODE") == "5"
    p = ex(_conv(256, 96, 27, 256, 3, 2, 1), tile="96x256x16x1x8x1x1x32x2x2")          # strided: the table gather
    assert p.startswith("bodahip_conv_big_f32 96x256x16_w1x8_p2_big") and mode(p, "J_MODE") == "2"
    p = ex(_conv(256, 3, 227, 96, 11, 4, 0))                      # AlexNet conv1 (strided, unpadded, wide): row-decimated LDS patch (round 4), 33 row sets of 1 x 11 kernels
    assert p.startswith("bodahip_conv_big_f32 96x256x22_w1x8_big") and mode(p, "J_MODE") == "7" and "-DRDEC=1" in p and "-DKH0=11" in p and "-DSY0=4" in p and "-DCH=55" in p and "-DPF=1" in p   # (round 6: of the staging-wave kernel)
    os.environ["BODAHIP_CBIG"] = "off"
    try: q = ex(_conv(256, 3, 227, 96, 11, 4, 0)); assert q.startswith("bodahip_conv_f32 32x256x22_w1x4") and "-DRDEC=1" in q
    finally: del os.environ["BODAHIP_CBIG"]
    p = ex(_conv(64, 3, 224, 64, 7, 2, 3))                        # GoogLeNet / ResNet conv1 (padded): row gather (KW >= 6)
    assert mode(p, "J_MODE") == "6" and "-DJROWS=" in p
    p = ex(_conv(128, 256, 27, 256, 1))                           # NiN cccp3 at 128 images: 1x1, tiled kernel (K = 256 is not "short")
    assert p.startswith("bodahip_conv_f32 ") and mode(p, "J_MODE") == "5"
    p = ex(_conv(256, 256, 27, 256, 1))                           # ... at 256 images (round 4c): the 16-bytes-per-lane streaming kernel, eight waves x 64 out_chans
    assert p.startswith("bodahip_k1_quad_f32 64x1024x256_w1x8") and "-DOCB=2" in p and "-DMINW=2" in p
    p = ex(_conv(256, 256, 6, 4096, 6))                           # AlexNet fc6: 256 tiles of 64 x 64 -> the fully-connected kernel (round 4): eight waves, three LDS stages
    assert p.startswith("bodahip_fc_f32 64x64x64_w2x4_m16_p2") and "-DTM=64" in p and "-DTN=64" in p
    p = ex(_conv(256, 4096, 1, 1000, 1))                          # AlexNet fc8: 64 tiles of 64 x 64 would starve the chip -> 256 tiles of 32 x 32
    assert p.startswith("bodahip_fc_f32 32x32x64_w2x4_m16_p2") and "-DTM=32" in p and "-DTN=32" in p
    p = ex(_conv(128, 256, 6, 4096, 6))                           # fc6 at 128 images: 64 images x 32 out_chans
    assert "-DTM=64" in p and "-DTN=32" in p
    p = ex(_conv(5, 256, 6, 4096, 6))                             # ... at 5 images: the tiled kernel's thin 16x16-MFMA tiles, contiguous images
    assert p.startswith("bodahip_conv_f32 ") and mode(p, "J_MODE") in ("3", "4")
    p = ex(_conv(256, 96, 27, 256, 3, 2, 1))                      # stride 2 in x: per-element table gather
    assert mode(p, "J_MODE") == "2"
    p = ex(_conv(64, 64, 56, 256, 1))                             # ResNet-50 res2 64 -> 256 at B=64: the streaming 1x1 kernel, one out_chan tile
    assert p.startswith("bodahip_k1_stream_f32 256x64x64_w8x1") and "-DKC=64" in p and "-DHW=3136" in p and "-DOCB=1" in p and "-DCB=2" in p and "-DEDGE_OC=0" in p
    assert ex(_conv(4, 64, 56, 256, 1)).startswith("bodahip_conv_f32 ")            # ... but not at small batch
    p = ex(_conv(256, 96, 55, 96, 1))                             # NiN cccp1: the 16-bytes-per-lane streaming kernel (round 4), one workgroup of four waves per CU, ring of 8 K steps
    assert p.startswith("bodahip_k1_quad_f32 96x512x96_w1x4") and "-DKC=96" in p and "-DHW=3025" in p and "-DOCB=3" in p and "-DRING=8" in p and "-DMINW=1" in p and "-DEDGE_OC=0" in p
    assert ex(_conv(256, 96, 6, 96, 1)).startswith("bodahip_conv_f32 ")                       # ... only on long pel axes
    assert ex(_conv(256, 96, 55, 96, 1), tile="64x64x16x2x2x2").startswith("bodahip_conv_f32 64x64x16_w2x2")  # an explicit tile wins
    # bf16: channel-innermost LDS patch for stride-1-in-x kernels on >= 16 channels (a multiple of 8), the gather kernel otherwise;
    # split-K only for tile-starved long-K shapes (fc6), never for the big layers
    p = ex(_conv(256, 256, 13, 384, 3, 1, 1, func="hip_conv_bf16"))
    assert p.startswith("bodahip_conv_patch_bf16 128x128x144_w2x2") and "-DCG=2" in p
    assert ex(_conv(256, 3, 227, 96, 11, 4, 0, func="hip_conv_bf16")).startswith("s2d(48x57x57,k3x3)+bodahip_conv_patch_bf16 96x128x144_w1x4")   # conv1: space-to-depth front end
    assert ex(_conv(64, 3, 224, 64, 7, 2, 3, func="hip_conv_bf16")).startswith("s2d(16x115x115,k4x4)+bodahip_conv_patch_bf16 64x")               # 7x7/2 pad 3 -> 4x4 on 16 channels
    assert ex(_conv(64, 96, 27, 256, 3, 2, 1, func="hip_conv_bf16")).startswith("bodahip_conv_bf16 ")                                            # strided on many channels: gather kernel
    p = ex(_conv(256, 256, 6, 4096, 6, func="hip_conv_bf16"))
    assert p.startswith("bodahip_conv_bf16 ") and "_s" in p.split()[1] and "-DSPLITK=1" in p
    assert "-DSPLITK" not in ex(_conv(64, 1024, 14, 256, 1, func="hip_conv_bf16"))


def test_planner_round5_rules_k1_tiles_and_the_k_hand_off_tile_field():
    """Round 5, host logic only: (1) a 1x1 fp32 layer whose 128 x 128 tiles still make >= 3.5 rounds of the CUs takes them (in-sequence A/B: NiN cccp5 at 256 images,
    cccp3 at 128), with fewer tiles the 64 x 64 tiles stay; (2) the eleventh tile field asks for sequential K hand-off (-DKHO=1): segments are lowered until none is
    empty, it is an fp32 convolution form (ignored by sgemm), and it is refused together with K slices or staging waves."""
    ex = R.explain_plan
    os.environ["BODAHIP_CBIG"] = "off"   # (the tiled kernel's own rules; round 6 hands most of these layers to the staging-wave kernel: next test)
    try:
        assert ex(_conv(256, 384, 13, 384, 1)).startswith("bodahip_conv_f32 128x128x16_w2x2 ")        # cccp5 @256: 3 x 338 = 1014 tiles
        assert ex(_conv(128, 256, 27, 256, 1)).startswith("bodahip_conv_f32 128x128x16_w2x2 ")        # cccp3 @128: 2 x 729 = 1458 tiles
        assert ex(_conv(128, 384, 13, 384, 1)).startswith("bodahip_conv_f32 64x64x16_w2x2_p2 ")       # cccp5 @128: 507 tiles
        assert ex(_conv(256, 1024, 6, 1024, 1)).startswith("bodahip_conv_f32 64x64x32_w2x2_p2 ")      # cccp7 @256: 576 tiles
        assert ex(_conv(256, 1024, 6, 1000, 1)).startswith("bodahip_conv_f32 64x64x32_w2x2_p2 ")      # cccp8: out_chans not a multiple of 128
    finally: del os.environ["BODAHIP_CBIG"]
    p = ex(_conv(128, 384, 6, 1024, 3, 1, 1), tile="128x128x36x2x2x2x1x32x1x0x8")
    assert p.startswith("bodahip_conv_f32 128x128x36_w2x2_h8 ") and "-DKHO=1" in p and "-DJ_MODE=7" in p
    p = ex(_conv(2, 16, 9, 40, 3, 1, 1), tile="64x64x16x2x2x2x1x32x1x0x64")                       # K = 144 = 8 steps of 18: at most 8 segments
    assert "_h8 " in p and "-DKHO=1" in p
    p = ex(_conv(2, 8, 9, 40, 3, 1, 1), tile="64x64x72x2x2x2x1x32x1x0x4")                          # one K step: no hand-off at all
    assert "_h" not in p.split(" ")[1] and "-DKHO" not in p
    assert "-DKHO" not in ex(_conv(8, 64, 13, 128, 1), tile="64x64x16x2x2x2x2x32x1x0x3")          # with K slices (a re-associating form): the hand-off is dropped
    with pytest.raises(UnsupErr, match="unsupported tile configuration"):                         # with staging waves (they leave the kernel early): refused
        ex(_conv(8, 64, 13, 128, 1), tile="64x64x16x2x2x1x1x32x1x1x3")
    sg = lambda m, n, k: __import__("boda_amd.op", fromlist=["parse_op"]).parse_op(f"(str_vals=(type=sgemm),nda_vals=(a=(dims=(K={k},M={m})),b=(dims=(K={k},N={n})),c=(dims=(M={m},N={n}))))")
    assert "-DKHO" not in ex(sg(512, 512, 512), tile="64x64x16x2x2x2x1x32x1x0x4")


def test_planner_k1_chain_and_the_net_driver_fuses_nin_block_one():
    """hip_conv_k1_chain (round 4c): two chained 1x1 convolutions plan onto the chain form of the 16-bytes-per-lane streaming kernel (both filter images in LDS, the
    intermediate channels = the rows of one accumulator set), and ConvPipeFwd fuses exactly NiN's cccp1 -> cccp2 at the bench batches (dry init, no device)."""
    from boda_amd.cnn_op import OpTune, add_codegen_annotations as _aca, annotate_k1_chain, k1_chain_applies
    def add_codegen_annotations(op, tune):   # (_conv() ops arrive annotated: strip, annotate again)
        op = op.copy(); op.str_vals.pop("func_name"); op.nda_vals.pop("conv_has_relu"); return _aca(op, tune)
    a = add_codegen_annotations(_conv(128, 96, 55, 96, 1), OpTune()); b = add_codegen_annotations(_conv(128, 96, 55, 96, 1), OpTune())
    ch = annotate_k1_chain(a, b, 1, 0)
    p = R.explain_plan(ch)
    assert p.startswith("bodahip_k1_chain_f32 96x512x96_w1x4") and "-DCHAIN=1" in p and "-DMID=96" in p and "-DOCB2=3" in p and "-DRELU=1" in p and "-DRELU2=0" in p and "-DRING=8" in p
    ch2 = annotate_k1_chain(add_codegen_annotations(_conv(4, 33, 20, 64, 1), OpTune()), add_codegen_annotations(_conv(4, 64, 20, 100, 1), OpTune()), 1, 1)
    p2 = R.explain_plan(ch2)
    assert "-DKC=33" in p2 and "-DMID=64" in p2 and "-DOCB=2" in p2 and "-DOCB2=4" in p2 and "-DEDGE_OC2=1" in p2 and "-DRING=1" in p2     # 17 K steps: a ring of one
    assert not k1_chain_applies(add_codegen_annotations(_conv(4, 256, 27, 256, 1), OpTune()), add_codegen_annotations(_conv(4, 256, 27, 256, 1), OpTune()))   # cccp3 -> cccp4
    from boda_amd.conv_pipe import ConvPipeFwd, DryRtc, alexnet_ng_conv, nin_imagenet
    for batch, want in ((128, [("cccp1", "cccp2")]), (2, [])):
        f = ConvPipeFwd(DryRtc()); f.init(nin_imagenet(batch))
        assert f.k1_chains == want and (not want or [c.tag for c in f.fwd_calls][:3] == ["conv1", "cccp1+cccp2", "pool0"]) and list(f._lazy) == [a_ for a_, _ in want] and not f.fused_pools
        # (round 5, opt-in: pool0 / pool2 formed inside conv2 / conv3's patch loads -- the third call of the pass is conv2 then, and the pooled nodes are lazy ones too)
        f = ConvPipeFwd(DryRtc(), fuse_f32_pools=True); f.init(nin_imagenet(batch))
        assert not want or ([c.tag for c in f.fwd_calls][:3] == ["conv1", "cccp1+cccp2", "conv2"] and f.fused_pools == {"pool0": "conv2", "pool2": "conv3"} and list(f._lazy) == ["cccp1", "pool0", "pool2"])
    f = ConvPipeFwd(DryRtc(), fuse_k1_chains=False); f.init(nin_imagenet(128)); assert not f.k1_chains
    f = ConvPipeFwd(DryRtc()); f.init(alexnet_ng_conv(256)); assert not f.k1_chains       # fc6 -> fc7 -> fc8 are whole-input windows / too wide


def test_planner_sgemm_tiles_follow_problem_size():
    from boda_amd.op import parse_op
    def sg(M, N, K, fn="hip_sgemm"):
        return parse_op(f"(str_vals=(type=sgemm,func_name={fn}),nda_vals=(a=(dims=(K={K},M={M})),b=(dims=(K={K},N={N})),c=(dims=(M={M},N={N}))))")
    big, small = R.explain_plan(sg(8192, 8192, 8192)), R.explain_plan(sg(256, 256, 256))
    # round 4: eight multiplying + four staging waves; round 5: as 256 x 128 tiles (4 x 2 multiplying waves of 64 x 64) where those deal out in whole rounds (2048 tiles = 8 x 256 CUs)
    assert big.startswith("bodahip_sgemm_big_f32 256x128x8_w3x4_p4") and "-DBKS=8" in big and "-DPF=4" in big and "-DTBJ=128" in big and "-DWI=4" in big
    os.environ["BODAHIP_NO_SGEMM_256X128"] = "1"
    try: assert R.explain_plan(sg(8192, 8192, 8192)).startswith("bodahip_sgemm_big_f32 256x256x8_w3x4_p4")
    finally: del os.environ["BODAHIP_NO_SGEMM_256X128"]
    # ... and, asked for by their tile strings ("...x3x4": the twelve waves), its 128 x 128 (two workgroups per CU) and 128 x 256 forms
    assert R.explain_plan(sg(2048, 2048, 2048), tile="128x128x8x3x4x2").startswith("bodahip_sgemm_big_f32 128x128x8_w3x4_p4") and "-DMINW=2" in R.explain_plan(sg(2048, 2048, 2048), tile="128x128x8x3x4x2")
    assert "-DTBI=128 -DTBJ=256 -DWI=2 -DWJ=4" in R.explain_plan(sg(4096, 4096, 4096), tile="128x256x8x3x4x1")
    assert small.startswith("bodahip_sgemm_f32 ") and small.split()[1] != big.split()[1]
    assert "-DI_MODE=1" in R.explain_plan(sg(130, 64, 50)) and "-DJ_MODE=1" in R.explain_plan(sg(128, 66, 50))     # scalar staging for ragged M / N
    assert R.explain_plan(sg(8192, 8192, 8192), tile="128x128x16x2x2x2x4").count("-DSPLITK=1") == 1           # split-K only as an explicit tune
    assert R.explain_plan(sg(4096, 4096, 4096, "hip_sgemm_bf16")).startswith("bodahip_sgemm_bf16 256x256")
    # two-level tiling: 7168^3 is 784 tiles of 256x256 = 3 rounds of 256 CUs + 16 -> 27 tile rows (756 tiles) on the large tile, the last 256 rows on small
    # tiles; 8192^3 (1024 = 4 rounds exactly) is not split; an explicit tile or BODAHIP_NO_SGEMM_SPLIT switches it off
    sp = R.explain_plan(sg(7168, 7168, 7168))
    assert sp.startswith("rows<6912:256x256x8_w3x4_p4+rest:bodahip_sgemm_big_f32 64x64x16_w2x2_p2_stg"), sp   # (round 6: the rest on the staging-wave kernel's own 64 x 64 form)
    assert R.explain_plan(sg(6144, 6144, 6144)).startswith("rows<5376:") and not R.explain_plan(sg(12288, 12288, 12288)).startswith("rows<")
    # round 6: guillotine cuts into whole rounds -- 10240^3 (3200 tiles of 256 x 128 = 6.25 rounds of 512: two such workgroups share a CU) = rows < 8192 (2560) + the last
    # 2048 rows' first 8192 columns (512) + a 2048^2 corner on 64 x 64 tiles (1024 = one round of four per CU); 5120^3 = a strip on 64 x 64 + 4096^2 (512) + a strip
    q10 = R.explain_plan(sg(10240, 10240, 10240))
    assert q10.startswith("parts=3 [0+8192,0+10240]:256x128x8_w3x4_p4 [8192+2048,0+8192]:256x128x8_w3x4_p4 [8192+2048,8192+2048]:64x64x16_w2x2_p2_stg last:bodahip_sgemm_big_f32 64x64x16_w2x2_p2_stg "), q10
    assert R.explain_plan(sg(5120, 5120, 5120)).startswith("parts=3 ") and "[1024+4096,0+4096]:256x128x8_w3x4_p4" in R.explain_plan(sg(5120, 5120, 5120))
    os.environ["BODAHIP_NO_SGEMM_PARTS"] = "1"
    try: assert R.explain_plan(sg(10240, 10240, 10240)).startswith("bodahip_sgemm_big_f32 256x128x8") and R.explain_plan(sg(5120, 5120, 5120)).startswith("rows<3072:")   # (round 5's plans)
    finally: del os.environ["BODAHIP_NO_SGEMM_PARTS"]
    assert R.explain_plan(sg(7168, 7168, 7168), tile="128x128x16x2x2x2").startswith("bodahip_sgemm_f32 128x128")
    assert not R.explain_plan(sg(7168, 7170, 7168)).startswith("rows<")        # ragged N: scalar staging, no split
    os.environ["BODAHIP_NO_SGEMM_SPLIT"] = "1"
    try: assert R.explain_plan(sg(7168, 7168, 7168)).startswith("bodahip_sgemm_f32 ")
    finally: del os.environ["BODAHIP_NO_SGEMM_SPLIT"]
    # round 6: the sizes the general kernel would give 64 x 64 tiles (768^3 .. 3072^3 of the list) run the staging-wave kernel's 64 x 64 form -- four multiplying waves of
    # one block, four staging waves (512 threads), register budget for four workgroups per CU; ragged M / N and short K stay on the general kernel; BODAHIP_NO_SGEMM_STG64
    for n in (1024, 2048, 3072):
        q = R.explain_plan(sg(n, n, n)); assert q.startswith("bodahip_sgemm_big_f32 64x64x16_w2x2_p2_stg ") and "-DWI=2 -DWJ=2 -DMINW=4" in q, q
    assert R.explain_plan(sg(1024, 1026, 1024)).startswith("bodahip_sgemm_f32 ") and R.explain_plan(sg(1024, 1024, 256)).startswith("bodahip_sgemm_f32 ")
    os.environ["BODAHIP_NO_SGEMM_STG64"] = "1"
    try: assert R.explain_plan(sg(2048, 2048, 2048)).startswith("bodahip_sgemm_f32 64x64x32_w2x2_p2 ")
    finally: del os.environ["BODAHIP_NO_SGEMM_STG64"]
    with pytest.raises(Exception): R.explain_plan(sg(2048, 2048, 2048), tile="64x64x8x2x2x4x1x32x2x3")     # (half a float4 unit per staging thread: refused on the host, not by the compiler)
    # the documented switch BODAHIP_SGEMM_BIG=off leaves every size on the general kernel (round-5 advisor finding: the wide-tile rule used to hand it the staging-wave kernel's x3x4 tile)
    os.environ["BODAHIP_SGEMM_BIG"] = "off"
    try:
        for n in (2048, 4096, 8192, 12288): assert R.explain_plan(sg(n, n, n)).startswith("bodahip_sgemm_f32 "), R.explain_plan(sg(n, n, n))
    finally: del os.environ["BODAHIP_SGEMM_BIG"]
    # the tile heuristic balances tiles over the CUs it is told about: a 512^3 problem on 256 CUs takes the thin 16x16-MFMA tiles, on 16 CUs 64x64
    assert R.explain_plan(sg(512, 512, 512), num_cus=256).split()[1] == "32x32x32_w2x2_m16_p8"
    assert R.explain_plan(sg(512, 512, 512), num_cus=16).split()[1] == "64x64x16_w2x2_p2_stg"


def test_prebuild_resolves_the_algorithm_like_conv_does_in_tolerance_mode():
    """op_tune hip_exact=0: conv() sends 3x3 / stride-1 layers with >= 96 input channels to the F(2x2,3x3) pipeline, so what prebuild compiles ahead of
    time and reports must be that pipeline's kernels -- not a direct kernel that never runs (a plan compiled at first use cannot be captured into a graph)."""
    from boda_amd.cnn_op import OpTune, add_codegen_annotations
    conv3 = _conv(256, 256, 13, 384, 3, 1, 1); conv3.str_vals.pop("func_name"); conv3.nda_vals.pop("conv_has_relu")
    tol = R.explain_plan(add_codegen_annotations(conv3, OpTune(hip_exact=0)))
    assert tol.startswith("winograd(F2x2,3x3)+bodahip_sgemm_f32 ") and "-DEPI=0" in tol
    assert R.explain_plan(add_codegen_annotations(conv3, OpTune())).startswith("bodahip_conv_big_f32 ")                   # bit-exact default: the direct (round 6: staging-wave) kernel
    res2 = _conv(64, 64, 56, 64, 3, 1, 1); res2.str_vals.pop("func_name"); res2.nda_vals.pop("conv_has_relu")
    assert R.explain_plan(add_codegen_annotations(res2, OpTune(hip_exact=0))).startswith("bodahip_conv_big_f32 ")        # too few channels for Winograd to pay: a direct kernel
    assert R.prebuild(add_codegen_annotations(conv3, OpTune(hip_exact=0))) > 4000                                          # and it cross-compiles
