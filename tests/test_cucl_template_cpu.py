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
    assert s.rstrip().endswith("}") and "+ 7U;" in s and "%(" not in s
    assert rtc.compile_offline(s) > 0                                         # compiles for gfx950 behind the CUCL prelude
    # a by-value scalar without a value stays an argument reference
    op2 = Op({"type": "own", "func_name": "own"}, dict(op.nda_vals, shift=Nda(None, "uint32_t", None)))
    assert "+ shift;" in instantiate(t, op2, "own__t1").src
    with pytest.raises(RtErr):
        instantiate(t, Op({"type": "own", "func_name": "own"}, {k: v for k, v in op.nda_vals.items() if k != "stride"}), "x")
    with pytest.raises(RtErr):
        instantiate(t, Op({"type": "own", "func_name": "own"}, dict(op.nda_vals, out=Nda(Dims.make("float", img=2, y=5, x=4)))), "x")
    with pytest.raises(UnsupErr):
        parse_template("dyn", "void f( GASQ float * const a ) // CUCL OUT_DYN x\n{}")


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


@pytest.mark.skipif(not os.path.isdir(REF_RTC), reason="no Boda checkout here (the GPU box has none)")
@pytest.mark.parametrize("name", sorted(REF_OPS))
def test_reference_generic_templates_instantiate_and_compile_for_gfx950(name):
    t = load_template(REF_RTC, name)
    inst = instantiate(t, Op({"type": name, "func_name": name}, REF_OPS[name]), f"{name}__gen0")
    assert inst.arg_names == [a.vn for a in t.arg_decls] and inst.blks >= 1 and "%(" not in inst.src
    assert rtc.compile_offline(inst.src) > 0
