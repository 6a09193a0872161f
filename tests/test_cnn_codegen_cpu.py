"""SURVEY.md section 8 F4, host side (no GPU): the restated blocking / variant annotations (src/cnn_op.cc, src/gbt_tile.H) against the launch
geometries the survey probed from the reference's own fixtures, and -- where a Boda checkout is present (the build container) -- the
reference's templates instantiated through the restated custom code generation (src/cnn_codegen.cc) and compiled for gfx950."""
import os
import pytest

import bench
from boda_amd import rtc
from oracle import cnn_codegen as cc
from boda_amd.cnn_op import OpTune
from boda_amd.op import RtErr, UnsupErr

REF_RTC = "/root/reference/test/rtc"
have_ref = pytest.mark.skipif(not os.path.isdir(REF_RTC), reason="no Boda checkout on this machine (the templates are not part of this repository)")
KT = OpTune(k1conv=1, tconv=1)


def test_good_div_and_gbt_tile():
    assert cc.good_div(7, 8) == 7 and cc.good_div(64, 8) == 8 and cc.good_div(55, 8) == 8    # 55 -> 7 chunks of 8: 1 of 56 wasted
    assert cc.good_div(13, 8) == 7                                                             # 2 x 8 wastes 3 of 13 (>= 20 %): 2 x 7
    g = cc.GbtTile((8, 8), 128, (256 * 55 * 55, 96))                                           # NiN cccp1 at batch 256
    assert g.mn_per_thr == (8, 8) and g.thr_per_blk == (10, 12) and g.num_blk == (9680, 1)


def test_annotations_reproduce_the_reference_geometries():
    """variant + blocks of AlexNet and NiN layers at batch 256 with k1conv=1, tconv=1; sgemm blocks."""
    al = [cc.annotate_ref(op, KT) for op in bench.alexnet_b256_ops()]
    assert [a.get_func_name() for a in al] == ["tconv"] * 5 + ["conv"] * 3
    w = al[0].get_dims("work")
    assert (w.dsz("blk_y"), w.dsz("out_chan_tile")) == (10, 12) and w.dsz("blk_bline") * w.dsz("blk_bx") * w.dsz("out_chan_blk") == 9856
    blks = lambda a: (a.get_dims("work").dsz("blk_bline") * a.get_dims("work").dsz("blk_bx") if a.get_func_name() == "tconv" else a.get_dims("work").dsz("pels_blk")) * a.get_dims("work").dsz("out_chan_blk")
    assert [blks(a) for a in al] == [9856, 6912, 2496, 2496, 1664, 128, 128, 32]
    nin = bench.nin_ops()
    k1 = [cc.annotate_ref(nin[i], KT) for i in (1, 4, 7, 10)]
    assert all(a.get_func_name() == "k1conv" for a in k1) and [blks(a) for a in k1] == [9680, 5832, 2028, 1152]
    assert k1[0].get_dims("in").names == ("blk", "blk_iter", "blk_iter_chan", "blk_pel") and k1[0].get_dims("in_ref") == nin[1].get_dims("in")
    assert k1[0].get_dims("filts").names == ("out_chan_blk", "in_chan", "y", "x", "out_chan_reg", "out_chan_tile")
    # without the enables every conv is the general variant; 1x1 with padding falls back to it too (src/cnn_op.cc:51-53)
    assert cc.annotate_ref(nin[1], OpTune()).get_func_name() == "conv"
    sg = {op.sgemm_geom()["M"]: op for op in bench.sgemm_full_ops()}
    w = cc.annotate_ref(sg[8192], OpTune()).get_dims("work")
    assert (w.dsz("Mg"), w.dsz("Ng"), w.dsz("Mb"), w.dsz("Nb"), w.dsz("Kb"), w.dsz("Mt"), w.dsz("Nt")) == (128, 64, 8, 16, 8, 8, 8)
    with pytest.raises(RtErr):          # the reference's default tune cannot run 64^3 (N = 64 is not a multiple of 128: src/cnn_op.cc:352-355)
        cc.annotate_ref(sg[64], OpTune())
    with pytest.raises(UnsupErr):
        cc.annotate_ref(nin[1], OpTune(k1conv=1, use_local_mem=2))     # vw defaults to 8: an OpenCL vector type, HIP has float2 / float4
    with pytest.raises(RtErr):
        cc.annotate_ref(bench.alexnet_b256_ops()[1], OpTune(use_local_mem=2, vw=4))     # conv_simd: Kb must be 1 (the reference asserts it, src/cnn_op.cc:246)
    cs = cc.annotate_ref(bench.alexnet_b256_ops(4)[0], OpTune(use_local_mem=2, vw=4, Kb=1))     # conv1 11x11 / 4: planes padded 227 -> 228, outputs on the 57x57 grid
    assert cs.get_func_name() == "conv_simd" and cs.get_dims("in_pels").sizes == (4, 228, 228) and cs.get_dims("out_pels").sizes == (4, 57, 57)
    assert cs.get_dims("in").names == ("chan", "pel") and cs.get_dims("in").dsz("pel") >= 4 * 228 * 228 + 7 * 228 + 7 and cs.get_dims("in").dsz("pel") % 4 == 0
    ks = cc.annotate_ref(nin[4], OpTune(k1conv=1, tconv=1, use_local_mem=2, vw=4))      # k1conv_simd: in / filts / out as (chan, pel) matrices padded to the blocking
    w = ks.get_dims("work")
    assert ks.get_func_name() == "k1conv_simd" and ks.get_dims("in").names == ("chan", "pel") and ks.get_dims("out").names == ("chan", "pel")
    assert ks.get_dims("out").sizes == (w.dsz("out_chan_blk") * w.dsz("out_chan_tile") * w.dsz("out_chan"), w.dsz("pels_blk") * w.dsz("pels_tile") * w.dsz("pels"))
    assert [x[0] for x in cc.xpose_ops(ks)] == ["k1conv_simd_xpose_filts", "k1conv_simd_xpose_in"] and [x[0] for x in cc.post_xpose_ops(ks)] == ["k1conv_simd_xpose_out"]
    assert [cc.annotate_ref(sg[2048], OpTune(use_local_mem=lm, vw=4)).get_func_name() for lm in (0, 1, 2, 3)] == ["sgemm_no_local", "sgemm", "sgemm_simd", "sgemm_simd_local"]
    # ipconv (1x1 output, no padding: fc6 at 256 images): the (pels, out_chan) blocking of conv plus fioc_tile lanes over the reduction -- the largest
    # power of two <= 32 that keeps the block at <= 512 threads (src/cnn_op.cc:204-209); in / filts stay in the reference layout, no layout pass
    ip = cc.annotate_ref(bench.alexnet_b256_ops()[5], OpTune(k1conv=1, tconv=1, ipconv=1))
    w = ip.get_dims("work")
    assert ip.get_func_name() == "ipconv" and w.names[-1] == "fioc_tile" and w.dsz("fioc_tile") * w.dsz("pels_tile") * w.dsz("out_chan_tile") <= 512
    assert w.dsz("fioc_tile") in (4, 8, 16, 32) and ip.get_dims("filts").names == ("out_chan", "in_chan", "y", "x") and cc.xpose_ops(ip) == []


@have_ref
@pytest.mark.parametrize("which", ["sgemm", "sgemm_no_local", "sgemm_simd", "sgemm_simd_local", "conv", "conv_simd", "k1conv", "k1conv_simd", "tconv", "ipconv"])
def test_reference_templates_instantiate_and_compile(which):
    sg2048 = [o for o in bench.sgemm_full_ops() if o.sgemm_geom()["M"] == 2048][0]
    op, tune = {"sgemm": (sg2048, OpTune()), "sgemm_no_local": (sg2048, OpTune(use_local_mem=0)), "sgemm_simd": (sg2048, OpTune(use_local_mem=2, vw=4)),
                "sgemm_simd_local": (sg2048, OpTune(use_local_mem=3, vw=4)),
                "conv": (bench.alexnet_b256_ops(4)[5], KT), "k1conv": (bench.nin_ops(4)[4], KT), "tconv": (bench.alexnet_b256_ops(4)[1], KT), "conv_simd": (bench.alexnet_b256_ops(4)[1], OpTune(use_local_mem=2, vw=4, Kb=1)), "k1conv_simd": (bench.nin_ops(4)[4], OpTune(k1conv=1, tconv=1, use_local_mem=2, vw=4)),
                "ipconv": (bench.alexnet_b256_ops(4)[6], OpTune(k1conv=1, tconv=1, ipconv=1))}[which]
    anno = cc.annotate_ref(op, tune)
    assert anno.get_func_name() == which
    inst = cc.instantiate_ref(REF_RTC, which, anno, "t_" + which)
    assert inst.tpb > 0 and inst.blks > 0 and "%(" not in inst.src
    assert rtc.compile_offline(inst.src, use_cache=False) > 0
    for tname, src, dst, xop in cc.xpose_ops(anno) + cc.post_xpose_ops(anno):
        xi = cc.instantiate_ref(REF_RTC, tname, xop, "t_x_" + tname)
        assert rtc.compile_offline(xi.src, use_cache=False) > 0


@have_ref
def test_build_recipe_writes_code_objects_and_manifest():
    from oracle import ref_cucl
    import json
    if not os.path.exists(os.path.join(ref_cucl.OUT, "manifest.json")):
        assert ref_cucl.build() > 0
    man = json.load(open(os.path.join(ref_cucl.OUT, "manifest.json")))
    assert len(man) >= 30 and all(os.path.getsize(os.path.join(ref_cucl.OUT, (e.get("main") or e["l1"]["main"])["file"])) > 1000 for e in man)
    assert {e["variant"] for e in man} >= {"sgemm", "conv", "k1conv", "tconv", "k1conv_chain"}


def test_k1conv_write_xposed_chain_annotation():
    """enable_write_xpose (src/rtc_fwd.cc:495-503): the first k1conv's `out` takes the second's `in` dims; the padded out_chans / pel blocks of the two layers agree."""
    ops = bench.nin_ops(20)[1:3]
    a1, a2 = cc.annotate_ref(ops[0], KT), cc.annotate_ref(ops[1], KT)
    cc.chain_k1conv(a1, a2)
    o = a1.get_dims("out")
    assert o == a2.get_dims("in") and o.names == ("blk", "blk_iter", "blk_iter_chan", "blk_pel") and o.sizes == (757, 12, 8, 80)      # (cccp1 at 20 images: blk=757, blk_pel=80)
    assert a1.get_dims("out_ref") == ops[0].get_dims("out")
    with pytest.raises(UnsupErr):
        cc.chain_k1conv(cc.annotate_ref(bench.nin_ops(20)[0], KT), a2)           # conv1 is a tconv, not a k1conv
    with pytest.raises(UnsupErr):
        cc.annotate_ref(bench.sgemm_full_ops()[5], OpTune(prof_variant=1))        # sgemm_prof: not instantiable (in the reference either)


@have_ref
def test_reference_reduce_template_multi_pack():
    """test/rtc/reduce.cucl: `float_multi ins` + gen_op_reduce (src/cnn_codegen.cc:28-34) -> one pointer argument and one accumulation line per member."""
    from boda_amd.op import Dims, Nda, Op
    d = Dims.make("float", img=2, chan=3, y=4, x=5)
    vals = {"out": Nda(d), "ins_num": Nda(Dims((), (), "uint32_t"), "uint32_t", (4,))}
    vals.update({f"ins_{i}": Nda(d) for i in range(4)})
    inst = cc.instantiate_ref(REF_RTC, "reduce", Op({"type": "Reduce", "func_name": "reduce"}, vals), "t_reduce")
    assert inst.arg_names == ["ins_num", "ins_0", "ins_1", "ins_2", "ins_3", "out"] and inst.tpb == 256 and inst.blks == 1
    assert inst.src.count("[GLOB_ID_1D];") == 4 and "%(" not in inst.src
    assert rtc.compile_offline(inst.src, use_cache=False) > 0
