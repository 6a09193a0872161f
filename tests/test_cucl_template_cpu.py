"""CUCL template instantiation (boda_amd/cucl_template.py, the generic half of src/rtc_func_gen.cc): template variables, index
expressions and launch geometry on an own template; and -- where a Boda checkout is present (this container, not the GPU box) --
every static generic template of the reference's test/rtc instantiated for a sample op and compiled for gfx950 by the
backend's hiprtc path, i.e. source compatibility of the CUCL dialect under be=hip."""
import os
import pytest

from boda_amd.cucl_template import instantiate, load_template, parse_template
from boda_amd.op import Dims, Nda, Op, RtErr, UnsupErr
from boda_amd import rtc

OWN = """
CUCL_GLOBAL_KERNEL void %(rtc_func_name)( GASQ float const * const in, // CUCL IN img:chan:y:x
                                          uint32_t const shift, // CUCL IN :
                                          GASQ void const * const stride, // CUCL REF y:x
                                          GASQ float * const out ) // CUCL OUT img:chan:y:x
{
  // CUCL IX GLOB_ID_1D out
  // CUCL IX pel in use_dims=img:y:x
  if( GLOB_ID_1D >= %(GLOB_ID_1D_dims_prod) ) { return; }
  int32_t const iy = %(GLOB_ID_1D_y)*%(stride_y_dim), ix = %(GLOB_ID_1D_x)*%(stride_x_dim);
  out[GLOB_ID_1D] = in[%(GLOB_ID_1D_img)*%(in_img_stride) + %(GLOB_ID_1D_chan)*%(in_chan_stride) + iy*%(in_y_stride) + ix*%(in_x_stride)] + %(shift);
}
"""


def _nda4(b, c, y, x):
    return Nda(Dims.make("float", img=b, chan=c, y=y, x=x))


def _none_yx(y, x):
    return Nda(Dims(("y", "x"), (y, x), "none"), "none")


def test_own_template_vars_geometry_and_offline_compile():
    t = parse_template("own", OWN)
    assert [a.vn for a in t.arg_decls] == ["in", "shift", "stride", "out"] and [a.io_type for a in t.arg_decls] == ["IN", "IN", "REF", "OUT"]
    assert t.arg_decls[1].loi == 0 and t.arg_decls[0].loi == 1 and t.arg_decls[2].tn == "none"
    op = Op({"type": "own", "func_name": "own"}, {"in": _nda4(2, 3, 9, 7), "out": _nda4(2, 3, 5, 4), "stride": _none_yx(2, 2),
                                                  "shift": Nda(None, "uint32_t", (7,))})
    inst = instantiate(t, op, "own__t0")
    assert inst.arg_names == ["in", "shift", "stride", "out"] and inst.tpb == 256 and inst.blks == 1   # ceil(120 / 256)
    s = inst.src
    assert "void own__t0(" in s and "GLOB_ID_1D >= 120" in s
    assert "((GLOB_ID_1D/4)%5)*2" in s and "(GLOB_ID_1D%4)*2" in s           # y, x index expressions (stride 4 / 1) times the REF's dims
    assert "(GLOB_ID_1D/60)*189" in s and "((GLOB_ID_1D/20)%3)*63" in s      # outermost dim not wrapped; in strides 189 / 63
    assert s.rstrip().endswith("}") and "+ 7;" in s and "%(" not in s
    assert rtc.compile_offline(s) > 0                                         # compiles for gfx950 behind the CUCL prelude
    # a by-value scalar without a value stays an argument reference
    op2 = Op({"type": "own", "func_name": "own"}, dict(op.nda_vals, shift=Nda(None, "uint32_t", None)))
    assert "+ shift;" in instantiate(t, op2, "own__t1").src
    with pytest.raises(RtErr):
        instantiate(t, Op({"type": "own", "func_name": "own"}, {k: v for k, v in op.nda_vals.items() if k != "stride"}), "x")
    with pytest.raises(RtErr):
        instantiate(t, Op({"type": "own", "func_name": "own"}, dict(op.nda_vals, out=Nda(Dims.make("float", img=2, y=5, x=4)))), "x")
    with pytest.raises(RtErr):
        parse_template("dyn", "void f( float const a ) // CUCL IN_DYN :\n{}")      # by-value arguments must not be DYN


MULTI = """
CUCL_GLOBAL_KERNEL void %(rtc_func_name)(
#if 0
  GASQ float_multi const * const ins, // CUCL IN img:chan:y:x
#endif
  %(ins_decl)
  float const scale, // CUCL IN :
  GASQ float * const out ) // CUCL OUT img:chan:y:x
{
  // CUCL IX GLOB_ID_1D out
  if( GLOB_ID_1D >= %(out_dims_prod) ) { return; }
  float v = 0;
  %(ins_ops);
  out[GLOB_ID_1D] = v * scale + %(ins_2_chan_dim);
}
"""


def _multi_op(n):
    d = Dims.make("float", img=2, chan=3, y=4, x=5)
    vals = {"out": Nda(d), "scale": Nda(None, "float", None), "ins_num": Nda(Dims((), (), "uint32_t"), "uint32_t", (n,))}
    vals.update({f"ins_{i}": Nda(d) for i in range(n)})
    return Op({"type": "own_multi", "func_name": "own_multi"}, vals)


def _multi_hook(cg, name):
    for vn in cg.multi_args["ins"]:
        cg.line("ins_ops", f"v += {vn}[GLOB_ID_1D];")


def test_multi_argument_pack_expands_to_num_arguments():
    """`<type>_multi <vn>`: op[<vn>_num] arguments <vn>_0 .. (arg_decl_t::set_vn_tn / multi_iter, src/rtc_func_gen.cc:24-41,143-151,388-391): one
    declaration line each in %(<vn>_decl), their dims as template variables, the custom hook sees the expanded names."""
    t = parse_template("own_multi", MULTI)
    assert [(a.vn, a.tn, a.multi) for a in t.arg_decls if a.vn == "ins"] == [("ins", "float", True)]
    inst = instantiate(t, _multi_op(3), "own_multi__n3", custom=_multi_hook)
    assert inst.arg_names == ["ins_0", "ins_1", "ins_2", "scale", "out"]      # (the pack expands at its DECLARATION's position in the arg list)
    for i in range(3):
        assert f"GASQ float const * const ins_{i}," in inst.src and f"v += ins_{i}[GLOB_ID_1D];" in inst.src
    assert "+ 3;" in inst.src and inst.blks == 1 and "%(" not in inst.src
    assert rtc.compile_offline(inst.src) > 0
    with pytest.raises(RtErr, match="ins_num"):
        instantiate(t, Op({"type": "own_multi", "func_name": "own_multi"}, {k: v for k, v in _multi_op(3).nda_vals.items() if k != "ins_num"}), "x", custom=lambda cg, n: None)
    with pytest.raises(RtErr, match="ins_2"):
        instantiate(t, Op({"type": "own_multi", "func_name": "own_multi"}, {k: v for k, v in _multi_op(3).nda_vals.items() if k != "ins_2"}), "x", custom=_multi_hook)


DYN = """
CUCL_GLOBAL_KERNEL void %(rtc_func_name)( GASQ float * const a, // CUCL OUT_DYN K:M
                                          float const vi // CUCL IN :
                                          %(cucl_arg_info_decls) )
{
  // CUCL IX GLOB_ID_1D a
  if( GLOB_ID_1D >= %(a_dims_prod) ) { return; }
  a[GLOB_ID_1D] = %(vi) + %(GLOB_ID_1D_K)*1000 + %(GLOB_ID_1D_M) + %(a_M_dim);
}
"""


def test_dyn_argument_dims_become_trailing_cai_args():
    t = parse_template("dyn", DYN)
    op = Op({"type": "dyn", "func_name": "dyn"}, {"a": Nda(Dims.make("float", K=0, M=0)), "vi": Nda(None, "float", None)})
    inst = instantiate(t, op, "dyn__0")
    assert inst.arg_names == ["a", "vi", "cai__GLOB_ID_1D_K_dim", "cai__GLOB_ID_1D_K_stride", "cai__GLOB_ID_1D_M_dim", "cai__GLOB_ID_1D_M_stride",
                              "cai__GLOB_ID_1D_dims_prod", "cai__a_K_dim", "cai__a_K_stride", "cai__a_M_dim", "cai__a_M_stride", "cai__a_dims_prod"]
    assert inst.blks == 0 and inst.tpb == 256          # geometry is per call
    assert "GLOB_ID_1D >= cai__a_dims_prod" in inst.src and ",int32_t cai__a_M_stride" in inst.src and "%(" not in inst.src
    vals, tpb, blks = inst.call_args({"a": Dims.make("float", K=37, M=20)})
    assert vals["cai__a_K_stride"] == 20 and vals["cai__GLOB_ID_1D_dims_prod"] == 740 and (tpb, blks) == (256, 3)
    assert rtc.compile_offline(inst.src) > 0


REF_RTC = "/root/reference/test/rtc"
u32 = lambda v: Nda(None, "uint32_t", (v,))
f32 = lambda v: Nda(None, "float", (v,))
REF_OPS = {
    "relu": {"inout": _nda4(2, 8, 6, 6)},
    "copy": {"in": _nda4(2, 8, 6, 6), "ocix": u32(4), "out": _nda4(2, 24, 6, 6)},
    "split_copy": {"in": _nda4(2, 24, 6, 6), "icix": u32(4), "out": _nda4(2, 8, 6, 6)},
    "softmax": {"in": _nda4(2, 10, 3, 3), "prob": _nda4(2, 10, 3, 3)},
    "dropout": {"inout": _nda4(2, 8, 6, 6), "dropout_ratio": f32(0.5), "det_drop_seed": u32(3)},
    "ZeroIfNonPos": {"in": _nda4(2, 8, 6, 6), "cond": _nda4(2, 8, 6, 6), "out": _nda4(2, 8, 6, 6)},
    "pool": {"avg_pool": u32(0), "emit_out_in_yx": u32(0), "in": _nda4(2, 8, 13, 13), "kern_sz": _none_yx(3, 3), "stride": _none_yx(2, 2),
             "in_pad": _none_yx(0, 0), "out": _nda4(2, 8, 6, 6), "out_in_yx": _nda4(2, 8, 6, 6)},
    "lrn": {"alpha": f32(1e-4), "beta": f32(0.75), "k": f32(1.0), "local_size": u32(5), "in": _nda4(2, 8, 6, 6), "out": _nda4(2, 8, 6, 6),
            "emit_out_scale_base": u32(0), "out_scale_base": _nda4(2, 8, 6, 6)},
    "spreading": {"avg_pool": u32(0), "out": _nda4(2, 8, 6, 6), "out_grad_loss": _nda4(2, 8, 6, 6), "out_in_yx": _nda4(2, 8, 6, 6),
                  "kern_sz": _none_yx(3, 3), "stride": _none_yx(2, 2), "in_pad": _none_yx(0, 0), "in_grad_loss": _nda4(2, 8, 13, 13)},
    "sum_loss_over_imgs": {"loss_per_pel": Nda(Dims.make("float", img=4, y=3, x=3)), "loss": Nda(Dims.make("float", y=3, x=3))},
    "xpose_filts": {"filts_ref": Nda(Dims.make("float", out_chan=96, in_chan=3, y=11, x=11)),
                    "filts": Nda(Dims.make("float", out_chan_blk=1, in_chan=3, y=11, x=11, out_chan_reg=8, out_chan_tile=12))},
}


_dynf = lambda **d: Nda(Dims.make("float", **d))
REF_OPS.update({
    "gen_data_sgemm_a": {"a": _dynf(K=0, M=0), "mode": Nda(None, "uint32_t", None), "vi": Nda(None, "float", None)},
    "gen_data_sgemm_b": {"b": _dynf(K=0, N=0), "mode": Nda(None, "uint32_t", None), "vi": Nda(None, "float", None)},
    "gen_data_Convolution_in": {"in": _dynf(img=0, chan=0, y=0, x=0), "mode": Nda(None, "uint32_t", None), "vi": Nda(None, "float", None)},
    "gen_data_Convolution_filts": {"filts": _dynf(out_chan=0, in_chan=0, y=0, x=0), "mode": Nda(None, "uint32_t", None), "vi": Nda(None, "float", None)},
    "gen_data_Convolution_biases": {"biases": _dynf(out_chan=0), "mode": Nda(None, "uint32_t", None), "vi": Nda(None, "float", None)},
    "quantize": {"out": _dynf(img=0, chan=0, y=0, x=0), "max_val": u32(255), "drop_mask": u32(3)},
})


@pytest.mark.skipif(not os.path.isdir(REF_RTC), reason="no Boda checkout here (the GPU box has none)")
@pytest.mark.parametrize("name", sorted(REF_OPS))
def test_reference_generic_templates_instantiate_and_compile_for_gfx950(name):
    t = load_template(REF_RTC, name)
    inst = instantiate(t, Op({"type": name, "func_name": name}, REF_OPS[name]), f"{name}__gen0")
    n_reg = len(t.arg_decls)
    assert inst.arg_names[:n_reg] == [a.vn for a in t.arg_decls] and all(a.startswith("cai__") for a in inst.arg_names[n_reg:])
    assert (inst.blks >= 1 or inst.dyn_vars) and "%(" not in inst.src
    assert rtc.compile_offline(inst.src) > 0
