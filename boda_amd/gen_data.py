"""Deterministic on-device test data (CUCL-dialect source handed to rtc.compile()).

Same values as the reference's gen_data kernels (test/rtc/gen-util.h:1-9, test/rtc/gen_data_sgemm_{a,b}.cucl,
test/rtc/gen_data_Convolution_{in,filts,biases}.cucl): mode 5 = vi + det_hash_rand(flat_ix + per-tensor constant),
mode>=100 for sgemm = exact-answer pattern (a = 1000*m + k, b = identity), modes 2/3/4 coordinate ramps / impulse.
The source is this project's own text in the CUCL macro vocabulary (GLOB_ID_1D, GASQ, CUCL_GLOBAL_KERNEL, ...), with
dims passed as by-value uint32 args -- it exercises the backend's generic source path exactly like the reference's
templates do.  Data is generated ON the device (as src/rtc_prof.cc:71-92 does), never uploaded.
"""
from __future__ import annotations
from typing import Dict, List, Tuple

from .op import Dims, Op
from .rtc import RtcArg, RtcFuncCall, RtcFuncInfo

HASH_CONSTS = {("sgemm", "a"): 12738732, ("sgemm", "b"): 12738732, ("Convolution", "in"): 234234567,
               ("Convolution", "filts"): 8753985, ("Convolution", "biases"): 39475612}

_UTIL = """
CUCL_DEVICE float det_hash_rand( uint32_t const rv ) {
  uint32_t h = rv;
  h ^= h >> 16; h *= 0x85ebca6b;
  h ^= h >> 13; h *= 0xc2b2ae35;
  h ^= h >> 16;
  return fmaf( (float)( h ), ( 10.0f / (float)( U32_MAX ) ), -5.0f );
}
"""

_SGEMM_T = """
// a: K:M
// m_off / M_glob: this var holds columns [m_off, m_off+M) of a global K:M_glob tensor (batch-axis shard); unsharded: 0 / M
CUCL_GLOBAL_KERNEL void gen_data_sgemm_a%(SFX)( GASQ %(TN) * const a, uint32_t const mode, float const vi, uint32_t const K, uint32_t const M,
                                          uint32_t const m_off, uint32_t const M_glob ) {
  // CUCL SHARD2 a size=M off=m_off
  // (what lets a multi-device backend run the function on an `a` split along M: per device M and m_off become the shard's -- csrc/hip_multi.cc)
  uint32_t fin_mode = mode; if( fin_mode >= 100 ) { fin_mode = fin_mode / 100; }
  if( GLOB_ID_1D >= K*M ) { return; }
  uint32_t const k = GLOB_ID_1D / M; uint32_t const m = GLOB_ID_1D % M + m_off;
  uint32_t const gix = k * M_glob + m; // flat index in the global tensor
  float val = vi;
  if( fin_mode == 2 ) { val += m; }
  if( fin_mode == 3 ) { val += k; }
  else if( fin_mode == 4 ) { if( (m==M_glob/2) && (k==K/2) ) { val += 1.0f; } }
  else if( fin_mode == 5 ) { val += det_hash_rand( gix + 12738732 ); }
  else if( fin_mode == 6 ) { val += m*1000 + k; }
  store_float_to_rp_%(RPTN)( val, GLOB_ID_1D, a );
}
// b: K:N
CUCL_GLOBAL_KERNEL void gen_data_sgemm_b%(SFX)( GASQ %(TN) * const b, uint32_t const mode, float const vi, uint32_t const K, uint32_t const N ) {
  if( GLOB_ID_1D >= K*N ) { return; }
  uint32_t const k = GLOB_ID_1D / N; uint32_t const n = GLOB_ID_1D % N;
  float val = vi;
  if( mode == 2 ) { val += n; }
  if( mode == 3 ) { val += k; }
  else if( mode == 4 ) { if( (n==N/2) && (k==K/2) ) { val += 1.0f; } }
  else if( mode == 5 ) { val += det_hash_rand( GLOB_ID_1D + 12738732 ); }
  else if( mode >= 100 ) { if( n==k ) { val += 1.0f; } }
  store_float_to_rp_%(RPTN)( val, GLOB_ID_1D, b );
}
"""
# the sgemm generators exist per storage type, as the reference's templates do (`GASQ %(a_tn) * const a` + store_float_to_rp_%(a_tn),
# test/rtc/gen_data_sgemm_a.cucl:2,20): float, and half for 16-bit-storage sgemms (test/sgemm-ops-debug-half.txt)
def _sgemm_src(sfx: str, tn: str, rptn: str) -> str:
    return _SGEMM_T.replace("%(SFX)", sfx).replace("%(TN)", tn).replace("%(RPTN)", rptn)


SRC = _UTIL + _sgemm_src("", "float", "float") + _sgemm_src("_half", "_Float16", "half") + """
// 4-D tensors ?:?:y:x (Convolution in / filts) and the 1-D biases; hc = per-tensor hash constant
// ix_off: flat-index offset of this var inside the global tensor (img-axis shard of `in`: img0*chan*y*x); unsharded: 0
// (the index declaration is what lets a multi-device backend run the function on a var sharded along img: csrc/hip_multi.cc)
CUCL_GLOBAL_KERNEL void gen_data_Convolution_4d( GASQ float * const t, uint32_t const mode, float const vi, uint32_t const sz,
                                                 uint32_t const Y, uint32_t const X, uint32_t const hc, uint32_t const ix_off ) {
  // CUCL IX GLOB_ID_1D t
  if( GLOB_ID_1D >= sz ) { return; }
  uint32_t const x = GLOB_ID_1D % X; uint32_t const y = ( GLOB_ID_1D / X ) % Y;
  float val = vi;
  if( mode == 2 ) { val += x; }
  if( mode == 3 ) { val += y; }
  else if( mode == 4 ) { if( (x==X/2) && (y==Y/2) ) { val += 1.0f; } }
  else if( mode == 5 ) { val += det_hash_rand( GLOB_ID_1D + ix_off + hc ); }
  t[GLOB_ID_1D] = val;
}
CUCL_GLOBAL_KERNEL void gen_data_Convolution_biases( GASQ float * const biases, uint32_t const mode, float const vi, uint32_t const sz ) {
  if( GLOB_ID_1D >= sz ) { return; }
  float val = vi;
  if( mode == 5 ) { val += det_hash_rand( GLOB_ID_1D + 39475612 ); }
  biases[GLOB_ID_1D] = val;
}
"""

FUNCS: Dict[str, List[str]] = {
    "gen_data_sgemm_a": ["a", "mode", "vi", "K", "M", "m_off", "M_glob"],
    "gen_data_sgemm_b": ["b", "mode", "vi", "K", "N"],
    "gen_data_sgemm_a_half": ["a", "mode", "vi", "K", "M", "m_off", "M_glob"],
    "gen_data_sgemm_b_half": ["b", "mode", "vi", "K", "N"],
    "gen_data_Convolution_4d": ["t", "mode", "vi", "sz", "Y", "X", "hc", "ix_off"],
    "gen_data_Convolution_biases": ["biases", "mode", "vi", "sz"],
}
TPB = 256


def func_infos() -> List[RtcFuncInfo]:
    """All four generator functions as ONE source block (one hiprtc compile, as a reference compile() batch)."""
    out = []
    for i, (fn, args) in enumerate(FUNCS.items()):
        op = Op({"type": "gen_data", "func_name": fn}, {})
        out.append(RtcFuncInfo(fn, SRC if i == 0 else "", args, op))
    return out


def gen_call(op_type: str, arg: str, vn: str, dims: Dims, mode: int, vi: float, shard_off: int = 0, shard_glob: int = 0) -> RtcFuncCall:
    """The call that fills var `vn` (dims `dims`) with the reference's test pattern for (op_type, arg).
    Batch-axis shards: for sgemm `a`, shard_off/shard_glob = first global column m and global M; for Convolution `in`,
    shard_off = first global image (the var then holds the matching slice of the global tensor's pattern)."""
    n = dims.dims_prod()
    u32 = lambda v: RtcArg.scalar(v, "uint32_t")
    base = {"mode": u32(mode), "vi": RtcArg.scalar(vi, "float")}
    if op_type == "sgemm" and arg in ("a", "b"):
        other = "M" if arg == "a" else "N"
        am = {arg: RtcArg.var(vn), **base, "K": u32(dims.dsz("K")), other: u32(dims.dsz(other))}
        if arg == "a":
            am["m_off"] = u32(shard_off); am["M_glob"] = u32(shard_glob or dims.dsz("M"))
        fn = "gen_data_sgemm_" + arg + ("_half" if dims.tn == "half" else "")
    elif op_type == "Convolution" and arg in ("in", "filts"):
        am = {"t": RtcArg.var(vn), **base, "sz": u32(n), "Y": u32(dims.dsz("y")), "X": u32(dims.dsz("x")),
              "hc": u32(HASH_CONSTS[(op_type, arg)]),
              "ix_off": u32((shard_off * (n // dims.dsz("img"))) if arg == "in" else 0)}
        fn = "gen_data_Convolution_4d"
    elif op_type == "Convolution" and arg == "biases":
        am = {"biases": RtcArg.var(vn), **base, "sz": u32(n)}
        fn = "gen_data_Convolution_biases"
    else:
        raise ValueError(f"no gen_data for {op_type}.{arg}")
    return RtcFuncCall(fn, am, tpb=TPB, blks=(n + TPB - 1) // TPB)
