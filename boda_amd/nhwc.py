"""Channels-last bf16 tensors for the bf16 convolution path (BASELINE config 5): annotation and layout passes.

The reference reaches its fast convolution variants through *transposed* operands: `add_cnn_codegen_annotations` gives the annotated op
new dims for `in` / `filts` / `out` and keeps the original ones as `<arg>_ref` (src/cnn_op.cc:142-330); the harness then generates
data in the reference layout, runs `<func>_xpose_<arg>` to fill the kernel's layout, times ONLY the main function, and transposes the
result back (`profile_rcg_call`, src/rtc_prof.cc:92-121); in a full net the producing kernel writes the consumer's format directly
(src/rtc_fwd.cc:229-243,495-503).  Its one reduced-precision precedent is 16-bit storage with fp32 math (`__tn__=half`,
test/sgemm-ops-debug-half.txt; src/cnn_codegen.cc:440-449).

be=hip's bf16 variant does exactly that for the layout the bf16 matrix cores want -- contraction index contiguous in both operands:
    in     img:y:x:chan           bfloat16   (chan padded to a multiple of 8 with zero channels)
    filts  out_chan:y:x:in_chan   bfloat16   (in_chan padded likewise)
    out    img:y:x:chan           bfloat16   (or float: op_tune hip_out=f32)
    biases out_chan               float
func_name `hip_conv_nhwc` (kernels/conv_nhwc_bf16.hip); the layout passes below are CUCL-dialect sources that go through the backend's
generic hiprtc path like the reference's own `xpose_*` kernels.
"""
from __future__ import annotations
from typing import Dict, List, Tuple

from .op import Dims, Nda, Op
from .rtc import RtcArg, RtcFuncCall, RtcFuncInfo

FUNC = "hip_conv_nhwc"
ARGS = (("filts", "IN"), ("biases", "IN"), ("in", "IN"), ("stride", "REF"), ("in_pad", "REF"), ("out", "OUT"))


def pad8(c: int) -> int:
    return (c + 7) // 8 * 8


def nhwc_dims(d: Dims, tn: str = "bfloat16", pad: bool = True) -> Dims:
    """img:chan:y:x -> img:y:x:chan (chan padded to a multiple of 8 when `pad`)."""
    c = d.dsz("chan")
    return Dims(("img", "y", "x", "chan"), (d.dsz("img"), d.dsz("y"), d.dsz("x"), pad8(c) if pad else c), tn)


def ohwi_dims(d: Dims, tn: str = "bfloat16") -> Dims:
    """out_chan:in_chan:y:x -> out_chan:y:x:in_chan (in_chan padded to a multiple of 8)."""
    return Dims(("out_chan", "y", "x", "in_chan"), (d.dsz("out_chan"), d.dsz("y"), d.dsz("x"), pad8(d.dsz("in_chan"))), tn)


def annotate(a: Op, out_tn: str = "bfloat16") -> None:
    """In place: the `hip_conv_nhwc` form of an annotated Convolution -- kernel dims for in / filts / out, the originals as <arg>_ref."""
    for an, conv in (("in", lambda d: nhwc_dims(d)), ("filts", ohwi_dims), ("out", lambda d: nhwc_dims(d, out_tn, pad=False))):
        ref = a.get_dims(an)
        a.nda_vals[an + "_ref"] = Nda(dims=ref, tn=ref.tn)
        a.nda_vals[an] = Nda(dims=conv(ref), tn=conv(ref).tn)
    a.set_func_name(FUNC)


# layout passes (one thread per element of the destination; sizes by value).  `__bf16` conversions round to nearest even.
XPOSE_SRC = """
// in_ref img:chan:y:x float -> in img:y:x:chan(padded) bf16
CUCL_GLOBAL_KERNEL void hip_conv_nhwc_xpose_in( GASQ float const * const in_ref, GASQ __bf16 * const in, uint32_t const n, uint32_t const C,
                                                 uint32_t const CP, uint32_t const HW ) {
  uint32_t const i = GLOB_ID_1D;
  if( i >= n ) { return; }
  uint32_t const c = i % CP, pel = ( i / CP ) % HW, img = i / ( CP * HW );
  in[i] = ( c < C ) ? (__bf16)in_ref[( img*C + c )*HW + pel] : (__bf16)0.0f;
}
// filts_ref out_chan:in_chan:y:x float -> filts out_chan:y:x:in_chan(padded) bf16
CUCL_GLOBAL_KERNEL void hip_conv_nhwc_xpose_filts( GASQ float const * const filts_ref, GASQ __bf16 * const filts, uint32_t const n, uint32_t const C,
                                                    uint32_t const CP, uint32_t const HW ) {
  uint32_t const i = GLOB_ID_1D;
  if( i >= n ) { return; }
  uint32_t const c = i % CP, tap = ( i / CP ) % HW, oc = i / ( CP * HW );
  filts[i] = ( c < C ) ? (__bf16)filts_ref[( oc*C + c )*HW + tap] : (__bf16)0.0f;
}
// out img:y:x:chan (bf16 / float) -> out_ref img:chan:y:x float
CUCL_GLOBAL_KERNEL void hip_conv_nhwc_xpose_out_bf16( GASQ __bf16 const * const out, GASQ float * const out_ref, uint32_t const n, uint32_t const C, uint32_t const HW ) {
  uint32_t const i = GLOB_ID_1D;
  if( i >= n ) { return; }
  uint32_t const pel = i % HW, c = ( i / HW ) % C, img = i / ( C * HW );
  out_ref[i] = (float)out[( img*HW + pel )*C + c];
}
CUCL_GLOBAL_KERNEL void hip_conv_nhwc_xpose_out_f32( GASQ float const * const out, GASQ float * const out_ref, uint32_t const n, uint32_t const C, uint32_t const HW ) {
  uint32_t const i = GLOB_ID_1D;
  if( i >= n ) { return; }
  uint32_t const pel = i % HW, c = ( i / HW ) % C, img = i / ( C * HW );
  out_ref[i] = out[( img*HW + pel )*C + c];
}
"""
XPOSE_FUNCS: Dict[str, List[str]] = {
    "hip_conv_nhwc_xpose_in": ["in_ref", "in", "n", "C", "CP", "HW"],
    "hip_conv_nhwc_xpose_filts": ["filts_ref", "filts", "n", "C", "CP", "HW"],
    "hip_conv_nhwc_xpose_out_bf16": ["out", "out_ref", "n", "C", "HW"],
    "hip_conv_nhwc_xpose_out_f32": ["out", "out_ref", "n", "C", "HW"],
}
_TPB = 256
_u32 = lambda v: RtcArg.scalar(int(v), "uint32_t")


def ensure_compiled(rtc) -> None:
    if getattr(rtc, "_nhwc_xpose_compiled", False):
        return
    infos = [RtcFuncInfo(fn, XPOSE_SRC if i == 0 else "", args, Op({"type": "xpose", "func_name": fn}, {})) for i, (fn, args) in enumerate(XPOSE_FUNCS.items())]
    rtc.compile(infos)
    rtc._nhwc_xpose_compiled = True


def xpose_call(arg: str, ref_vn: str, vn: str, ref_dims: Dims, dims: Dims) -> RtcFuncCall:
    """The layout pass between `<arg>_ref` (reference layout, float) and `<arg>` (kernel layout): in / filts forward, out backward."""
    if arg == "in":
        n = dims.dims_prod()
        am = {"in_ref": RtcArg.var(ref_vn), "in": RtcArg.var(vn), "n": _u32(n), "C": _u32(ref_dims.dsz("chan")), "CP": _u32(dims.dsz("chan")),
              "HW": _u32(dims.dsz("y") * dims.dsz("x"))}
        return RtcFuncCall("hip_conv_nhwc_xpose_in", am, tpb=_TPB, blks=(n + _TPB - 1) // _TPB)
    if arg == "filts":
        n = dims.dims_prod()
        am = {"filts_ref": RtcArg.var(ref_vn), "filts": RtcArg.var(vn), "n": _u32(n), "C": _u32(ref_dims.dsz("in_chan")), "CP": _u32(dims.dsz("in_chan")),
              "HW": _u32(dims.dsz("y") * dims.dsz("x"))}
        return RtcFuncCall("hip_conv_nhwc_xpose_filts", am, tpb=_TPB, blks=(n + _TPB - 1) // _TPB)
    if arg == "out":
        n = ref_dims.dims_prod()
        fn = "hip_conv_nhwc_xpose_out_f32" if dims.tn == "float" else "hip_conv_nhwc_xpose_out_bf16"
        am = {"out": RtcArg.var(vn), "out_ref": RtcArg.var(ref_vn), "n": _u32(n), "C": _u32(ref_dims.dsz("chan")), "HW": _u32(ref_dims.dsz("y") * ref_dims.dsz("x"))}
        return RtcFuncCall(fn, am, tpb=_TPB, blks=(n + _TPB - 1) // _TPB)
    raise ValueError(arg)


# ------------------------------------------------------------------------------------------------
# non-conv forward kernels on channels-last bf16 tensors (full-net driver, boda_amd/conv_pipe.py): semantics of test/rtc/{pool,lrn,relu,
# copy}.cucl, 8 channels (one 16-byte chunk) per thread.  Arithmetic in fp32, results rounded to bf16 once (RNE) when stored.
# ------------------------------------------------------------------------------------------------
FWD_SRC = """
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
// pooling: one thread per (img, oy, ox, 8 channels); the window is clipped to the plane (padding never takes part; an average divides by
// the clipped area), taps column by column as the reference sums an average (test/rtc/pool.cucl)
CUCL_GLOBAL_KERNEL void nhwc_pool( GASQ bf16x8_t const * const in, GASQ bf16x8_t * const out, uint32_t const n, uint32_t const C8, uint32_t const H,
                                   uint32_t const W, uint32_t const OH, uint32_t const OW, uint32_t const KH, uint32_t const KW, uint32_t const SY,
                                   uint32_t const SX, uint32_t const PY, uint32_t const PX, uint32_t const avg_pool ) {
  uint32_t const i = GLOB_ID_1D;
  if( i >= n ) { return; }
  uint32_t const c = i % C8, ox = ( i / C8 ) % OW, oy = ( i / ( C8*OW ) ) % OH, img = i / ( C8*OW*OH );
  int32_t const y0 = (int32_t)( oy*SY ) - (int32_t)PY, x0 = (int32_t)( ox*SX ) - (int32_t)PX;
  int32_t const ya = ( y0 < 0 ) ? 0 : y0, xa = ( x0 < 0 ) ? 0 : x0;
  int32_t const yb = ( y0 + (int32_t)KH > (int32_t)H ) ? (int32_t)H : y0 + (int32_t)KH, xb = ( x0 + (int32_t)KW > (int32_t)W ) ? (int32_t)W : x0 + (int32_t)KW;
  float acc[8];
  for( int32_t e = 0; e != 8; ++e ) { acc[e] = avg_pool ? 0.0f : -FLT_MAX; }
  for( int32_t x = xa; x < xb; ++x ) {
    for( int32_t y = ya; y < yb; ++y ) {
      bf16x8_t const v = in[( ( img*H + y )*W + x )*C8 + c];
      for( int32_t e = 0; e != 8; ++e ) { float const f = (float)v[e]; acc[e] = avg_pool ? ( acc[e] + f ) : ( ( f > acc[e] ) ? f : acc[e] ); }
    }
  }
  float const area = (float)( ( ( yb > ya ) ? yb - ya : 0 ) * ( ( xb > xa ) ? xb - xa : 0 ) );
  bf16x8_t r;
  for( int32_t e = 0; e != 8; ++e ) { r[e] = (__bf16)( avg_pool ? acc[e] / area : acc[e] ); }
  out[i] = r;
}
// across-channel LRN (test/rtc/lrn.cucl): out[c] = in[c] * ( k + alpha/local_size * sum_{|d| <= local_size/2} in[c+d]^2 ) ^ -beta; one thread per element
CUCL_GLOBAL_KERNEL void nhwc_lrn( GASQ __bf16 const * const in, GASQ __bf16 * const out, uint32_t const n, uint32_t const C, uint32_t const local_size,
                                  float const alpha, float const beta, float const k ) {
  uint32_t const i = GLOB_ID_1D;
  if( i >= n ) { return; }
  int32_t const c = i % C, half = local_size / 2;
  float sumsq = 0.0f;
  for( int32_t d = -half; d <= half; ++d ) {
    if( c + d >= 0 && c + d < (int32_t)C ) { float const v = (float)in[(int32_t)i + d]; sumsq += v*v; }
  }
  out[i] = (__bf16)( (float)in[i] * powf( k + sumsq * ( alpha / (float)local_size ), -beta ) );
}
// stand-alone ReLU (one that could not be fused into its conv)
CUCL_GLOBAL_KERNEL void nhwc_relu( GASQ __bf16 * const inout, uint32_t const n ) {
  uint32_t const i = GLOB_ID_1D;
  if( i < n ) { if( (float)inout[i] <= 0.0f ) { inout[i] = (__bf16)0.0f; } }
}
// Concat: copy one input (C8_in chunks per position) into its channel range of the output (src/rtc_fwd.cc:267-280)
CUCL_GLOBAL_KERNEL void nhwc_copy( GASQ bf16x8_t const * const in, GASQ bf16x8_t * const out, uint32_t const n, uint32_t const C8_in, uint32_t const C8_out,
                                   uint32_t const off8 ) {
  uint32_t const i = GLOB_ID_1D;
  if( i >= n ) { return; }
  uint32_t const pel = i / C8_in;
  out[pel*C8_out + off8 + ( i - pel*C8_in )] = in[i];
}
"""
FWD_FUNCS: Dict[str, List[str]] = {
    "nhwc_pool": ["in", "out", "n", "C8", "H", "W", "OH", "OW", "KH", "KW", "SY", "SX", "PY", "PX", "avg_pool"],
    "nhwc_lrn": ["in", "out", "n", "C", "local_size", "alpha", "beta", "k"],
    "nhwc_relu": ["inout", "n"],
    "nhwc_copy": ["in", "out", "n", "C8_in", "C8_out", "off8"],
}
_f32 = lambda v: RtcArg.scalar(float(v), "float")


def ensure_fwd_compiled(rtc) -> None:
    ensure_compiled(rtc)
    if getattr(rtc, "_nhwc_fwd_compiled", False):
        return
    infos = [RtcFuncInfo(fn, FWD_SRC if i == 0 else "", args, Op({"type": "fwd", "func_name": fn}, {})) for i, (fn, args) in enumerate(FWD_FUNCS.items())]
    rtc.compile(infos)
    rtc._nhwc_fwd_compiled = True


def pool_call(in_vn: str, out_vn: str, i: Dims, o: Dims, kern, stride, pad, avg: int) -> RtcFuncCall:
    """i / o: the channels-last dims of the vars (img:y:x:chan, chan a multiple of 8)."""
    c8 = i.dsz("chan") // 8; n = o.dsz("img") * o.dsz("y") * o.dsz("x") * c8
    am = {"in": RtcArg.var(in_vn), "out": RtcArg.var(out_vn), "n": _u32(n), "C8": _u32(c8), "H": _u32(i.dsz("y")), "W": _u32(i.dsz("x")), "OH": _u32(o.dsz("y")),
          "OW": _u32(o.dsz("x")), "KH": _u32(kern[0]), "KW": _u32(kern[1]), "SY": _u32(stride[0]), "SX": _u32(stride[1]), "PY": _u32(pad[0]), "PX": _u32(pad[1]),
          "avg_pool": _u32(avg)}
    return RtcFuncCall("nhwc_pool", am, tpb=_TPB, blks=(n + _TPB - 1) // _TPB)


def lrn_call(in_vn: str, out_vn: str, d: Dims, local_size: int, alpha: float, beta: float, k: float) -> RtcFuncCall:
    n = d.dims_prod()
    am = {"in": RtcArg.var(in_vn), "out": RtcArg.var(out_vn), "n": _u32(n), "C": _u32(d.dsz("chan")), "local_size": _u32(local_size), "alpha": _f32(alpha),
          "beta": _f32(beta), "k": _f32(k)}
    return RtcFuncCall("nhwc_lrn", am, tpb=_TPB, blks=(n + _TPB - 1) // _TPB)


def relu_call(vn: str, d: Dims) -> RtcFuncCall:
    n = d.dims_prod()
    return RtcFuncCall("nhwc_relu", {"inout": RtcArg.var(vn), "n": _u32(n)}, tpb=_TPB, blks=(n + _TPB - 1) // _TPB)


def copy_call(in_vn: str, out_vn: str, i: Dims, o: Dims, chan_off: int) -> RtcFuncCall:
    n = i.dims_prod() // 8
    am = {"in": RtcArg.var(in_vn), "out": RtcArg.var(out_vn), "n": _u32(n), "C8_in": _u32(i.dsz("chan") // 8), "C8_out": _u32(o.dsz("chan") // 8), "off8": _u32(chan_off // 8)}
    return RtcFuncCall("nhwc_copy", am, tpb=_TPB, blks=(n + _TPB - 1) // _TPB)
