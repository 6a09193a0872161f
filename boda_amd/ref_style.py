"""Reference-SHAPED sgemm and conv for the same silicon: what Boda's own code generator structure gives on MI355X.

The reference cannot be built here, so its generated kernels cannot be timed on this GPU.  This module
restates the *structure* of its default `sgemm` variant -- `op_tune` MNt=8:8, MNb=8:16, Kb=8 (`src/cnn_op.H:18-20`,
`src/cnn_op.cc:338-378`): 128 threads per workgroup, every thread an 8x8 register block of `c`, both operands staged through
local memory in K steps of 8, plain fp32 FMAs -- as our own CUCL-dialect source, and runs it through the backend's *generic*
path (hiprtc + 1-D launch, `CUCL_BACKEND_IX 3`), i.e. exactly where the reference's templates would run under `be=hip`.
It is a measurement baseline ("what a VALU register-tiled kernel of the reference's shape reaches on this chip") beside the
MFMA kernels, and a second GPU implementation for parity: each output is the same ascending-k fma chain.

Same operand contract as `sgemm` (`test/rtc/sgemm.cucl:1-4`): a K:M, b K:N, c M:N, row-major; M % 64 == N % 128 == K % 8 == 0
(the reference's default tune has the same divisibility rule, `src/cnn_op.cc:352-355`).
"""
from __future__ import annotations
from typing import Tuple

from .op import Dims, Op, UnsupErr
from .rtc import HipCompute, RtcArg, RtcFuncCall, RtcFuncInfo

FUNC = "ref_style_sgemm"
TPB = 128
SRC = """
// 8x16 threads, 8x8 outputs each -> 64 x 128 tile of c per workgroup; K step 8
CUCL_GLOBAL_KERNEL void ref_style_sgemm( GASQ float const * const a, GASQ float const * const b, GASQ float * const c,
                                         uint32_t const M, uint32_t const N, uint32_t const K ) {
  LOCSHAR_MEM float a_sm[8*64];
  LOCSHAR_MEM float b_sm[8*128];
  uint32_t const tid = LOC_ID_1D;
  uint32_t const tm = tid >> 4, tn = tid & 15;
  uint32_t const n_blks = N >> 7;
  uint32_t const bm = GRP_ID_1D / n_blks, bn = GRP_ID_1D % n_blks;
  float acc[8][8];
  for( uint32_t i = 0; i != 8; ++i ) { for( uint32_t j = 0; j != 8; ++j ) { acc[i][j] = 0.0f; } }
  GASQ float const * const a_blk = a + bm*64;
  GASQ float const * const b_blk = b + bn*128;
  for( uint32_t k0 = 0; k0 < K; k0 += 8 ) {
    for( uint32_t e = tid; e < 8*64; e += 128 ) { a_sm[e] = a_blk[( k0 + ( e >> 6 ) )*M + ( e & 63 )]; }
    for( uint32_t e = tid; e < 8*128; e += 128 ) { b_sm[e] = b_blk[( k0 + ( e >> 7 ) )*N + ( e & 127 )]; }
    BARRIER_SYNC;
    for( uint32_t kk = 0; kk != 8; ++kk ) {
      float a_r[8]; float b_r[8];
      for( uint32_t i = 0; i != 8; ++i ) { a_r[i] = a_sm[kk*64 + tm*8 + i]; }
      for( uint32_t j = 0; j != 8; ++j ) { b_r[j] = b_sm[kk*128 + j*16 + tn]; }   // column j*16+tn: conflict-free across tn
      for( uint32_t i = 0; i != 8; ++i ) { for( uint32_t j = 0; j != 8; ++j ) { acc[i][j] = fmaf( a_r[i], b_r[j], acc[i][j] ); } }
    }
    BARRIER_SYNC;
  }
  for( uint32_t i = 0; i != 8; ++i ) {
    for( uint32_t j = 0; j != 8; ++j ) { c[( bm*64 + tm*8 + i )*N + bn*128 + j*16 + tn] = acc[i][j]; }
  }
}
"""
ARGS = ["a", "b", "c", "M", "N", "K"]

# The same register-tiled structure for Convolution + bias + ReLU -- the shape of the reference's generic `conv` variant
# (`test/rtc/conv.cucl` + `gen_op_conv`, `src/cnn_codegen.cc:165-216`: filters and an input patch staged through local memory, an
# 8x8 block of (out_chan, pel) outputs per thread, K = in_chan*ky*kx walked in steps of 8).  NCHW / OIHW as `cudnn_conv`; any
# sizes (edges are range-tested); same ascending-k fma chain per output as the oracle.
CONV_FUNC = "ref_style_conv"
CONV_SRC = """
CUCL_GLOBAL_KERNEL void ref_style_conv( GASQ float const * const filts, GASQ float const * const biases, GASQ float const * const in,
                                        GASQ float * const out, uint32_t const B, uint32_t const C, uint32_t const H, uint32_t const W,
                                        uint32_t const OC, uint32_t const KH, uint32_t const KW, uint32_t const SY, uint32_t const SX,
                                        uint32_t const PY, uint32_t const PX, uint32_t const OH, uint32_t const OW, uint32_t const relu ) {
  LOCSHAR_MEM float a_sm[8*64];    // filts tile: 8 k x 64 out_chan
  LOCSHAR_MEM float b_sm[8*128];   // im2col tile: 8 k x 128 pels
  uint32_t const tid = LOC_ID_1D;
  uint32_t const tm = tid >> 4, tn = tid & 15;
  uint32_t const n_pel = B*OH*OW, K = C*KH*KW;
  uint32_t const n_blks = ( n_pel + 127 ) >> 7;
  uint32_t const bm = GRP_ID_1D / n_blks, bn = GRP_ID_1D % n_blks;
  float acc[8][8];
  for( uint32_t i = 0; i != 8; ++i ) { for( uint32_t j = 0; j != 8; ++j ) { acc[i][j] = 0.0f; } }
  for( uint32_t k0 = 0; k0 < K; k0 += 8 ) {
    for( uint32_t e = tid; e < 8*64; e += 128 ) {
      uint32_t const k = k0 + ( e & 7 ), oc = bm*64 + ( e >> 3 );
      a_sm[( e & 7 )*64 + ( e >> 3 )] = ( k < K && oc < OC ) ? filts[oc*K + k] : 0.0f;
    }
    for( uint32_t e = tid; e < 8*128; e += 128 ) {
      uint32_t const k = k0 + ( e >> 7 ), pel = bn*128 + ( e & 127 );
      float v = 0.0f;
      if( k < K && pel < n_pel ) {
        uint32_t const ic = k / ( KH*KW ), ky = ( k / KW ) % KH, kx = k % KW;
        uint32_t const img = pel / ( OH*OW ), oy = ( pel / OW ) % OH, ox = pel % OW;
        int32_t const iy = (int32_t)( oy*SY + ky ) - (int32_t)PY, ix = (int32_t)( ox*SX + kx ) - (int32_t)PX;
        if( iy >= 0 && ix >= 0 && iy < (int32_t)H && ix < (int32_t)W ) { v = in[( ( img*C + ic )*H + iy )*W + ix]; }
      }
      b_sm[e] = v;
    }
    BARRIER_SYNC;
    for( uint32_t kk = 0; kk != 8; ++kk ) {
      float a_r[8]; float b_r[8];
      for( uint32_t i = 0; i != 8; ++i ) { a_r[i] = a_sm[kk*64 + tm*8 + i]; }
      for( uint32_t j = 0; j != 8; ++j ) { b_r[j] = b_sm[kk*128 + j*16 + tn]; }
      for( uint32_t i = 0; i != 8; ++i ) { for( uint32_t j = 0; j != 8; ++j ) { acc[i][j] = fmaf( a_r[i], b_r[j], acc[i][j] ); } }
    }
    BARRIER_SYNC;
  }
  for( uint32_t i = 0; i != 8; ++i ) {
    uint32_t const oc = bm*64 + tm*8 + i;
    if( oc >= OC ) { continue; }
    for( uint32_t j = 0; j != 8; ++j ) {
      uint32_t const pel = bn*128 + j*16 + tn;
      if( pel >= n_pel ) { continue; }
      float v = acc[i][j] + biases[oc];
      if( relu ) { v = ( v > 0.0f ) ? v : 0.0f; }
      uint32_t const img = pel / ( OH*OW );
      out[( img*OC + oc )*( OH*OW ) + ( pel - img*( OH*OW ) )] = v;
    }
  }
}
"""
CONV_ARGS = ["filts", "biases", "in", "out", "B", "C", "H", "W", "OC", "KH", "KW", "SY", "SX", "PY", "PX", "OH", "OW", "relu"]


def compile_into(rtc: HipCompute) -> None:
    if not getattr(rtc, "_ref_style_compiled", False):
        rtc.compile([RtcFuncInfo(FUNC, SRC, ARGS, Op({"type": "sgemm", "func_name": FUNC}, {})),
                     RtcFuncInfo(CONV_FUNC, CONV_SRC, CONV_ARGS, Op({"type": "Convolution", "func_name": CONV_FUNC}, {}))])
        rtc._ref_style_compiled = True


def conv_call(filts_vn: str, biases_vn: str, in_vn: str, out_vn: str, g: dict, relu: bool = True) -> RtcFuncCall:
    """`g` = Op.conv_geom() of the Convolution."""
    u = lambda v: RtcArg.scalar(int(v), "uint32_t")
    am = {"filts": RtcArg.var(filts_vn), "biases": RtcArg.var(biases_vn), "in": RtcArg.var(in_vn), "out": RtcArg.var(out_vn), "relu": u(int(relu))}
    am.update({k: u(g[k]) for k in ("B", "C", "H", "W", "OC", "KH", "KW", "SY", "SX", "PY", "PX", "OH", "OW")})
    n_pel = g["B"] * g["OH"] * g["OW"]
    return RtcFuncCall(CONV_FUNC, am, tpb=TPB, blks=((g["OC"] + 63) // 64) * ((n_pel + 127) // 128))


def call(a_vn: str, b_vn: str, c_vn: str, M: int, N: int, K: int) -> RtcFuncCall:
    if M % 64 or N % 128 or K % 8:
        raise UnsupErr(f"ref_style_sgemm: M={M} N={N} K={K} must be multiples of 64 / 128 / 8 (the reference's default-tune rule)")
    u = lambda v: RtcArg.scalar(int(v), "uint32_t")
    return RtcFuncCall(FUNC, {"a": RtcArg.var(a_vn), "b": RtcArg.var(b_vn), "c": RtcArg.var(c_vn), "M": u(M), "N": u(N), "K": u(K)},
                       tpb=TPB, blks=(M // 64) * (N // 128))


def time_sgemm(rtc: HipCompute, size: int, iters: int = 5) -> Tuple[float, float]:
    """Square sgemm of `size` on gen_data mode-5 inputs; -> (best ms, TFLOP/s).  Creates and releases its own vars."""
    from . import gen_data as gd
    compile_into(rtc)
    if not getattr(rtc, "_gen_data_compiled", False):
        rtc.compile(gd.func_infos()); rtc._gen_data_compiled = True
    names = {}
    for an, d in (("a", Dims.make("float", K=size, M=size)), ("b", Dims.make("float", K=size, N=size)), ("c", Dims.make("float", M=size, N=size))):
        names[an] = f"refstyle_{an}_{size}"
        rtc.create_var_with_dims(names[an], d)
        if an != "c":
            rtc.run(gd.gen_call("sgemm", an, names[an], d, 5, 0.0))
    rfc = call(names["a"], names["b"], names["c"], size, size, size)
    ids = [rtc.run(rfc) for _ in range(iters + 1)]
    rtc.finish_and_sync()
    best = min(rtc.get_dur(i, i) for i in ids[1:])
    rtc.release_per_call_id_data()
    for vn in names.values():
        rtc.release_var(vn)
    return best, 2.0 * size ** 3 / (best * 1e-3) / 1e12


def time_conv(rtc: HipCompute, op: Op, iters: int = 3) -> Tuple[float, float]:
    """One Convolution op (gen_data mode-5 operands) through the reference-shaped conv kernel; -> (best ms, TFLOP/s)."""
    from . import gen_data as gd
    compile_into(rtc)
    if not getattr(rtc, "_gen_data_compiled", False):
        rtc.compile(gd.func_infos()); rtc._gen_data_compiled = True
    names = {}
    for an in ("filts", "biases", "in", "out"):
        names[an] = f"refstyle_conv_{an}"
        rtc.create_var_with_dims(names[an], op.get_dims(an))
        if an != "out":
            rtc.run(gd.gen_call("Convolution", an, names[an], op.get_dims(an), 5, 0.0))
    rfc = conv_call(names["filts"], names["biases"], names["in"], names["out"], op.conv_geom())
    ids = [rtc.run(rfc) for _ in range(iters + 1)]
    rtc.finish_and_sync()
    best = min(rtc.get_dur(i, i) for i in ids[1:])
    rtc.release_per_call_id_data()
    for vn in names.values():
        rtc.release_var(vn)
    return best, op.flops() / (best * 1e-3) / 1e12
