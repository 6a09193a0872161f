"""ctypes wrapper of oracle/boda_oracle.c (TEST INFRASTRUCTURE ONLY -- see that file's header).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libboda_oracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "boda_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s", "libboda_oracle.so"])
    return _SO


def _load():
    build()
    try:
        return C.CDLL(_SO)
    except OSError:
        build(force=True)
        return C.CDLL(_SO)


_lib = _load()
_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
u32, f32, u64 = C.c_uint32, C.c_float, C.c_uint64

_lib.bo_det_hash_rand.restype = f32; _lib.bo_det_hash_rand.argtypes = [u32]
_lib.bo_gen_data_sgemm_a.argtypes = [_f32p, u32, u32, u32, f32]
_lib.bo_gen_data_sgemm_b.argtypes = [_f32p, u32, u32, u32, f32]
_lib.bo_gen_data_conv_in.argtypes = [_f32p, u32, u32, u32, u32, u32, f32]
_lib.bo_gen_data_conv_filts.argtypes = [_f32p, u32, u32, u32, u32, u32, f32]
_lib.bo_gen_data_conv_biases.argtypes = [_f32p, u32, u32, f32]
_lib.bo_sgemm.argtypes = [_f32p, _f32p, _f32p, u32, u32, u32]
_lib.bo_sgemm_f64acc.argtypes = [_f32p, _f32p, _f32p, u32, u32, u32]
_lib.bo_conv_fwd.argtypes = [_f32p, _f32p, _f32p, _f32p] + [u32] * 14
_lib.bo_ssds_diff.restype = C.c_double
_lib.bo_ssds_diff.argtypes = [_f32p, _f32p, u64, C.POINTER(C.c_double)]
_lib.bo_digest_plan.restype = u32
_lib.bo_digest_plan.argtypes = [u64, _u32p, u32, u64, _u64p, _u64p, u32]
_lib.bo_digest_f32.argtypes = [_f32p, u64, _u64p, _u64p, u32, C.POINTER(f32), C.POINTER(f32), _f32p]
_lib.bo_num_threads.restype = C.c_int
_lib.bo_pool_fwd.argtypes = [_f32p, _f32p] + [u32] * 13
_lib.bo_relu.argtypes = [_f32p, u64]
_lib.bo_lrn_fwd.argtypes = [_f32p, _f32p, u32, u32, u32, u32, u32, f32, f32, f32]


def num_threads() -> int:
    return int(_lib.bo_num_threads())


def det_hash_rand(rv: int) -> float:
    return float(_lib.bo_det_hash_rand(rv & 0xFFFFFFFF))


def gen_sgemm_a(K, M, mode=5, vi=0.0):
    a = np.empty((K, M), np.float32); _lib.bo_gen_data_sgemm_a(a, K, M, mode, vi); return a


def gen_sgemm_b(K, N, mode=5, vi=0.0):
    b = np.empty((K, N), np.float32); _lib.bo_gen_data_sgemm_b(b, K, N, mode, vi); return b


def gen_conv_in(B, Cc, Y, X, mode=5, vi=0.0):
    t = np.empty((B, Cc, Y, X), np.float32); _lib.bo_gen_data_conv_in(t, B, Cc, Y, X, mode, vi); return t


def gen_conv_filts(OC, IC, Y, X, mode=5, vi=0.0):
    t = np.empty((OC, IC, Y, X), np.float32); _lib.bo_gen_data_conv_filts(t, OC, IC, Y, X, mode, vi); return t


def gen_conv_biases(OC, mode=5, vi=0.0):
    t = np.empty((OC,), np.float32); _lib.bo_gen_data_conv_biases(t, OC, mode, vi); return t


def sgemm(a: np.ndarray, b: np.ndarray, f64acc: bool = False) -> np.ndarray:
    """c[M,N] = sum_k a[k,M]*b[k,N]."""
    K, M = a.shape; K2, N = b.shape
    assert K == K2
    c = np.empty((M, N), np.float32)
    (_lib.bo_sgemm_f64acc if f64acc else _lib.bo_sgemm)(np.ascontiguousarray(a), np.ascontiguousarray(b), c, M, N, K)
    return c


def conv_fwd(inp, filts, biases, stride=(1, 1), pad=(0, 0), relu=True) -> np.ndarray:
    B, Cc, H, W = inp.shape; OC, IC, KH, KW = filts.shape
    assert IC == Cc
    SY, SX = stride; PY, PX = pad
    OH = (H + 2 * PY - KH) // SY + 1; OW = (W + 2 * PX - KW) // SX + 1
    out = np.empty((B, OC, OH, OW), np.float32)
    _lib.bo_conv_fwd(np.ascontiguousarray(inp), np.ascontiguousarray(filts), np.ascontiguousarray(biases), out,
                     B, Cc, H, W, OC, KH, KW, SY, SX, PY, PX, OH, OW, 1 if relu else 0)
    return out


def pool_out_sz(in_sz: int, k: int, s: int, p: int) -> int:
    """Caffe ceil convention (src/conv_util.cc:198-204)."""
    pin = in_sz + 2 * p
    return 1 if pin < k else -(-(pin - k) // s) + 1


def pool_fwd(inp, kern=(3, 3), stride=(2, 2), pad=(0, 0), avg=False) -> np.ndarray:
    B, Cc, H, W = inp.shape
    OH, OW = pool_out_sz(H, kern[0], stride[0], pad[0]), pool_out_sz(W, kern[1], stride[1], pad[1])
    out = np.empty((B, Cc, OH, OW), np.float32)
    _lib.bo_pool_fwd(np.ascontiguousarray(inp), out, B, Cc, H, W, kern[0], kern[1], stride[0], stride[1], pad[0], pad[1], OH, OW, 1 if avg else 0)
    return out


def relu(x) -> np.ndarray:
    y = np.ascontiguousarray(x, np.float32).copy(); _lib.bo_relu(y.reshape(-1), y.size); return y


def lrn_fwd(inp, local_size=5, alpha=1.0, beta=0.75, k=1.0) -> np.ndarray:
    B, Cc, H, W = inp.shape
    out = np.empty_like(inp, dtype=np.float32)
    _lib.bo_lrn_fwd(np.ascontiguousarray(inp), out, B, Cc, H, W, local_size, alpha, beta, k)
    return out


def to_bf16(x: np.ndarray) -> np.ndarray:
    """fp32 -> bf16 (round-to-nearest-even) -> fp32: what v_cvt_pk_bf16_f32 does to the operands of the bf16 kernels."""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u >> 16) & 1) + 0x7FFF
    return ((u + r) & 0xFFFF0000).astype(np.uint32).view(np.float32).reshape(x.shape)


def mrd(o1: np.ndarray, o2: np.ndarray) -> float:
    a = np.ascontiguousarray(o1, np.float32).reshape(-1); b = np.ascontiguousarray(o2, np.float32).reshape(-1)
    assert a.size == b.size
    st = (C.c_double * 5)()
    r = float(_lib.bo_ssds_diff(a, b, a.size, st))
    return float("nan") if st[4] else r


def digest(v: np.ndarray, dim_sizes, seed: int):
    """-> (min, max, samps[float32], strides[u64], offsets[u64]) with the reference's sample plan."""
    flat = np.ascontiguousarray(v, np.float32).reshape(-1)
    st, acc = [], 1
    for s in reversed(list(dim_sizes)):
        st.append(acc); acc *= s
    dstr = np.array(list(reversed(st)), np.uint32)
    assert acc == flat.size
    strides = np.zeros(4096, np.uint64); offsets = np.zeros(4096, np.uint64)
    ns = int(_lib.bo_digest_plan(flat.size, dstr, len(dstr), seed, strides, offsets, 4096))
    assert ns <= 4096
    samps = np.zeros(ns, np.float32); mn = f32(); mx = f32()
    _lib.bo_digest_f32(flat, flat.size, strides, offsets, ns, C.byref(mn), C.byref(mx), samps)
    return float(mn.value), float(mx.value), samps, strides[:ns].copy(), offsets[:ns].copy()


def run_op(op, mode: int = 5, vi: float = 0.0, relu: bool = True):
    """Generate the reference's deterministic inputs for `op` (a boda_amd.op.Op) and run it.  -> dict of arrays."""
    t = op.get_type()
    if t == "sgemm":
        g = op.sgemm_geom()
        a = gen_sgemm_a(g["K"], g["M"], mode, vi); b = gen_sgemm_b(g["K"], g["N"], mode, vi)
        return {"a": a, "b": b, "c": sgemm(a, b)}
    if t == "Convolution":
        g = op.conv_geom()
        i = gen_conv_in(g["B"], g["C"], g["H"], g["W"], mode, vi)
        f = gen_conv_filts(g["OC"], g["C"], g["KH"], g["KW"], mode, vi)
        bi = gen_conv_biases(g["OC"], mode, vi)
        return {"in": i, "filts": f, "biases": bi,
                "out": conv_fwd(i, f, bi, (g["SY"], g["SX"]), (g["PY"], g["PX"]), relu)}
    raise ValueError("oracle: unsupported op type " + t)
