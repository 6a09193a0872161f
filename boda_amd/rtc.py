"""ctypes mirror of the reference's `rtc_compute_t` (src/rtc_compute.H:35-97) over the C ABI of libbodahip.so.

Method names, argument meaning and error behaviour follow the reference interface so that harness code and tests read
like Boda's own callers (`rtc->create_var_with_dims(...)`, `rtc->compile(...)`, `rtc->run(rfc)`, ...).
There is NO fallback: if the HIP extension is missing or cannot be loaded this module raises at import.
"""
from __future__ import annotations
import ctypes as C
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np

from .op import Dims, Op, RtErr, UnsupErr, TYPE_SIZES

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "libbodahip.so")


def _load() -> C.CDLL:
    if not os.path.exists(SO_PATH):
        raise ImportError(f"boda_amd: HIP extension {SO_PATH} is missing -- run `python -m boda_amd.build` "
                          "(there is no CPU fallback for this backend)")
    return C.CDLL(SO_PATH)


_lib = _load()


class _CDims(C.Structure):
    _fields_ = [("tn", C.c_char_p), ("ndims", C.c_uint32), ("sizes", C.POINTER(C.c_uint32)), ("names", C.POINTER(C.c_char_p))]


class _CArg(C.Structure):
    _fields_ = [("name", C.c_char_p), ("kind", C.c_int32), ("var", C.c_char_p), ("dims", _CDims), ("data", C.c_void_p)]


class _CFuncInfo(C.Structure):
    _fields_ = [("func_name", C.c_char_p), ("func_src", C.c_char_p), ("n_args", C.c_uint32),
                ("arg_names", C.POINTER(C.c_char_p)), ("op", C.c_char_p)]


class _CCompileOpts(C.Structure):
    _fields_ = [("show_compile_log", C.c_uint32), ("enable_lineinfo", C.c_uint32), ("show_func_attrs", C.c_uint32),
                ("show_rtc_calls", C.c_uint32)]


_ctxp = C.c_void_p
ABI = {  # symbol -> (restype, argtypes); every symbol include/bodahip.h declares
    "bodahip_abi_version": (C.c_int, []),
    "bodahip_last_error": (C.c_char_p, []),
    "bodahip_create": (C.c_int, [C.POINTER(_ctxp), C.c_int]),
    "bodahip_create_be": (C.c_int, [C.POINTER(_ctxp), C.c_char_p, C.c_int]),
    "bodahip_create_multi": (C.c_int, [C.POINTER(_ctxp), C.c_uint32, C.POINTER(C.c_int)]),
    "bodahip_num_devices": (C.c_int, [_ctxp, C.POINTER(C.c_uint32)]),
    "bodahip_destroy": (None, [_ctxp]),
    "bodahip_set_gen_src": (C.c_int, [_ctxp, C.c_uint32, C.c_char_p]),
    "bodahip_init": (C.c_int, [_ctxp]),
    "bodahip_get_plat_tag": (C.c_int, [_ctxp, C.c_char_p, C.c_size_t]),
    "bodahip_create_var": (C.c_int, [_ctxp, C.c_char_p, C.POINTER(_CDims)]),
    "bodahip_create_view": (C.c_int, [_ctxp, C.c_char_p, C.POINTER(_CDims), C.c_char_p]),
    "bodahip_release_var": (C.c_int, [_ctxp, C.c_char_p]),
    "bodahip_get_var_dims": (C.c_int, [_ctxp, C.c_char_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_char_p, C.c_size_t]),
    "bodahip_set_var_to_zero": (C.c_int, [_ctxp, C.c_char_p]),
    "bodahip_compile": (C.c_int, [_ctxp, C.c_uint32, C.POINTER(_CFuncInfo), C.POINTER(_CCompileOpts)]),
    "bodahip_compile_code_object": (C.c_int, [_ctxp, C.c_void_p, C.c_size_t, C.c_uint32, C.POINTER(_CFuncInfo)]),
    "bodahip_compile_to_file": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int, C.c_char_p]),
    "bodahip_release_func": (C.c_int, [_ctxp, C.c_char_p]),
    "bodahip_release_all_funcs": (C.c_int, [_ctxp]),
    "bodahip_run": (C.c_int, [_ctxp, C.c_char_p, C.c_uint32, C.POINTER(_CArg), C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]),
    "bodahip_finish_and_sync": (C.c_int, [_ctxp]),
    "bodahip_release_per_call_id_data": (C.c_int, [_ctxp]),
    "bodahip_get_dur": (C.c_int, [_ctxp, C.c_uint32, C.c_uint32, C.POINTER(C.c_float)]),
    "bodahip_profile_start": (C.c_int, [_ctxp]),
    "bodahip_profile_stop": (C.c_int, [_ctxp]),
    "bodahip_copy_to_var": (C.c_int, [_ctxp, C.c_char_p, C.POINTER(_CDims), C.c_void_p]),
    "bodahip_copy_from_var": (C.c_int, [_ctxp, C.c_void_p, C.POINTER(_CDims), C.c_char_p]),
    "bodahip_get_raw_ptr": (C.c_int, [_ctxp, C.c_char_p, C.POINTER(C.c_void_p)]),
    "bodahip_graph_begin": (C.c_int, [_ctxp]),
    "bodahip_graph_end": (C.c_int, [_ctxp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "bodahip_graph_launch": (C.c_int, [_ctxp, C.c_uint32, C.POINTER(C.c_uint32)]),
    "bodahip_graph_end_deps": (C.c_int, [_ctxp, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "bodahip_graph_destroy": (C.c_int, [_ctxp, C.c_uint32]),
    "bodahip_get_stream": (C.c_int, [_ctxp, C.POINTER(C.c_void_p)]),
    "bodahip_compile_stats": (C.c_int, [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_double)]),
    "bodahip_get_device_info": (C.c_int, [_ctxp, C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "bodahip_set_tune": (C.c_int, [_ctxp, C.c_char_p, C.c_char_p]),
    "bodahip_last_launch": (C.c_int, [_ctxp, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                      C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "bodahip_compile_offline": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_size_t), C.c_char_p, C.c_size_t]),
    "bodahip_parse_op": (C.c_int, [C.c_char_p, C.c_char_p, C.c_size_t]),
    "bodahip_prebuild": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int, C.c_char_p, C.POINTER(C.c_size_t)]),
    "bodahip_explain_plan": (C.c_int, [C.c_char_p, C.c_int, C.c_char_p, C.c_char_p, C.c_size_t]),
}
for _n, (_r, _a) in ABI.items():
    _f = getattr(_lib, _n)  # AttributeError here == library does not export what the header declares
    _f.restype, _f.argtypes = _r, _a


def _chk(rc: int) -> None:
    if rc == 0:
        return
    msg = (_lib.bodahip_last_error() or b"").decode(errors="replace")
    if rc == 1:
        raise UnsupErr(msg)
    raise RtErr(msg)


def _cdims(d: Dims):
    sizes = (C.c_uint32 * max(1, len(d.sizes)))(*d.sizes)
    names = (C.c_char_p * max(1, len(d.names)))(*[n.encode() for n in d.names])
    cd = _CDims(d.tn.encode(), len(d.sizes), C.cast(sizes, C.POINTER(C.c_uint32)), C.cast(names, C.POINTER(C.c_char_p)))
    return cd, (sizes, names)  # keep-alives


_NP = {"float": np.float32, "double": np.float64, "int32_t": np.int32, "uint32_t": np.uint32, "uint16_t": np.uint16,
       "uint8_t": np.uint8, "half": np.float16, "bfloat16": np.uint16}   # (numpy has no bf16: the raw 16-bit patterns)


@dataclass
class RtcFuncInfo:
    """rtc_func_info_t (src/rtc_compute.H:23-28)."""
    func_name: str
    func_src: str
    arg_names: List[str]
    op: Op


@dataclass
class RtcArg:
    """rtc_arg_t (src/rtc_compute.H:103-115): var name, or value (dims + optional data; no data == REF/null)."""
    n: str = ""
    dims: Optional[Dims] = None
    v: Optional[np.ndarray] = None

    @staticmethod
    def var(name: str) -> "RtcArg":
        return RtcArg(n=name)

    @staticmethod
    def ref(dims: Dims) -> "RtcArg":
        return RtcArg(dims=dims)

    @staticmethod
    def scalar(value, tn: str) -> "RtcArg":
        return RtcArg(dims=Dims((), (), tn), v=np.array([value], dtype=_NP[tn]))

    def is_var(self) -> bool:
        return bool(self.n)


@dataclass
class RtcFuncCall:
    """rtc_func_call_t (src/rtc_compute.H:117-123).  The marshalled C form is cached on first run(); call invalidate()
    after mutating arg_map / tpb / blks of a call that has already been run."""
    rtc_func_name: str
    arg_map: Dict[str, RtcArg] = field(default_factory=dict)
    tpb: int = 0
    blks: int = 0

    def invalidate(self) -> None:
        self.__dict__.pop("_c_form", None)


class RtcCompileOpts:
    def __init__(self, show_compile_log=0, enable_lineinfo=0, show_func_attrs=0, show_rtc_calls=0):
        self.show_compile_log, self.enable_lineinfo = show_compile_log, enable_lineinfo
        self.show_func_attrs, self.show_rtc_calls = show_func_attrs, show_rtc_calls


class HipCompute:
    """`rtc_compute_t` with be=hip.  One instance per GPU (one process per GPU in multi-GPU runs)."""
    be = "hip"

    def __init__(self, device_ordinal: int = 0, be: str = "hip", devices: Optional[Sequence[int]] = None):
        self._ctx = _ctxp()
        self.be = be
        self.devices = list(devices) if devices else None
        if self.devices:   # N GPUs behind this one backend: vars sharded on img / M, weights replicated (include/bodahip.h: bodahip_create_multi)
            arr = (C.c_int * len(self.devices))(*self.devices)
            _chk(_lib.bodahip_create_multi(C.byref(self._ctx), len(self.devices), arr)); device_ordinal = self.devices[0]
        else:
            _chk(_lib.bodahip_create_be(C.byref(self._ctx), be.encode(), device_ordinal))
        self.device_ordinal = device_ordinal
        self._init_done = False

    def close(self):
        if getattr(self, "_ctx", None):
            _lib.bodahip_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- the 19 virtuals ----
    def init(self) -> None:
        _chk(_lib.bodahip_init(self._ctx))
        self._init_done = True

    def get_plat_tag(self) -> str:
        buf = C.create_string_buffer(512)
        _chk(_lib.bodahip_get_plat_tag(self._ctx, buf, 512))
        return buf.value.decode()

    def num_devices(self) -> int:
        """Devices behind this backend (1 for `(be=hip)` / `(be=cpu)`, N for `(be=hip,devices=...)`)."""
        n = C.c_uint32(0)
        _chk(_lib.bodahip_num_devices(self._ctx, C.byref(n)))
        return int(n.value)

    def create_var_with_dims(self, vn: str, dims: Dims) -> None:
        cd, ka = _cdims(dims)
        _chk(_lib.bodahip_create_var(self._ctx, vn.encode(), C.byref(cd)))

    def create_var_with_dims_as_reshaped_view_of_var(self, vn: str, dims: Dims, src_vn: str) -> None:
        cd, ka = _cdims(dims)
        _chk(_lib.bodahip_create_view(self._ctx, vn.encode(), C.byref(cd), src_vn.encode()))

    def release_var(self, vn: str) -> None:
        _chk(_lib.bodahip_release_var(self._ctx, vn.encode()))

    def get_var_dims(self, vn: str) -> Dims:
        tn = C.create_string_buffer(32); nd = C.c_uint32(16); sizes = (C.c_uint32 * 16)(); names = C.create_string_buffer(1024)
        _chk(_lib.bodahip_get_var_dims(self._ctx, vn.encode(), tn, 32, C.byref(nd), sizes, names, 1024))
        nm = names.raw.split(b"\0")[:nd.value]
        return Dims(tuple(x.decode() for x in nm), tuple(int(sizes[i]) for i in range(nd.value)), tn.value.decode())

    def set_var_to_zero(self, vn: str) -> None:
        _chk(_lib.bodahip_set_var_to_zero(self._ctx, vn.encode()))

    def compile(self, func_infos: Sequence[RtcFuncInfo], opts: Optional[RtcCompileOpts] = None) -> None:
        n = len(func_infos)
        arr = (_CFuncInfo * max(1, n))()
        keep = []
        for i, fi in enumerate(func_infos):
            an = (C.c_char_p * max(1, len(fi.arg_names)))(*[a.encode() for a in fi.arg_names])
            keep.append(an)
            arr[i] = _CFuncInfo(fi.func_name.encode(), fi.func_src.encode(), len(fi.arg_names), C.cast(an, C.POINTER(C.c_char_p)),
                                fi.op.to_str().encode())
        o = opts or RtcCompileOpts()
        co = _CCompileOpts(o.show_compile_log, o.enable_lineinfo, o.show_func_attrs, o.show_rtc_calls)
        _chk(_lib.bodahip_compile(self._ctx, n, arr, C.byref(co)))

    def compile_code_object(self, code: bytes, func_infos: Sequence[RtcFuncInfo]) -> None:
        """compile() for functions that arrive as a gfx950 code object (func_src of the infos is ignored)."""
        n = len(func_infos)
        arr = (_CFuncInfo * max(1, n))(); keep = []
        for i, fi in enumerate(func_infos):
            an = (C.c_char_p * max(1, len(fi.arg_names)))(*[a.encode() for a in fi.arg_names]); keep.append(an)
            arr[i] = _CFuncInfo(fi.func_name.encode(), b"", len(fi.arg_names), C.cast(an, C.POINTER(C.c_char_p)), fi.op.to_str().encode())
        buf = C.create_string_buffer(code, len(code))
        _chk(_lib.bodahip_compile_code_object(self._ctx, C.cast(buf, C.c_void_p), len(code), n, arr))

    def release_func(self, func_name: str) -> None:
        _chk(_lib.bodahip_release_func(self._ctx, func_name.encode()))

    def release_all_funcs(self) -> None:
        _chk(_lib.bodahip_release_all_funcs(self._ctx))

    def run(self, rfc: RtcFuncCall) -> int:
        cf = rfc.__dict__.get("_c_form")
        if cf is not None:  # hot path: one ctypes call, no Python-side allocation
            fn, n, arr, tpb, blks, _keep = cf
            cid = C.c_uint32()
            _chk(_lib.bodahip_run(self._ctx, fn, n, arr, tpb, blks, C.byref(cid)))
            return int(cid.value)
        n = len(rfc.arg_map)
        arr = (_CArg * max(1, n))()
        keep = []
        for i, (an, a) in enumerate(rfc.arg_map.items()):
            if a.is_var():
                arr[i] = _CArg(an.encode(), 0, a.n.encode(), _CDims(b"none", 0, None, None), None)
            else:
                if a.dims is None:
                    raise RtErr(f"run: arg {an!r} is neither a var nor a value")
                cd, ka = _cdims(a.dims); keep.append(ka)
                data = None
                if a.v is not None:
                    buf = np.ascontiguousarray(a.v); keep.append(buf)
                    if buf.nbytes != a.dims.bytes_sz():
                        raise RtErr(f"run: by-value arg {an!r}: {buf.nbytes} bytes != dims.bytes_sz() {a.dims.bytes_sz()}")
                    data = buf.ctypes.data
                arr[i] = _CArg(an.encode(), 1, None, cd, data)
        cid = C.c_uint32()
        fn = rfc.rtc_func_name.encode()
        _chk(_lib.bodahip_run(self._ctx, fn, n, arr, rfc.tpb, rfc.blks, C.byref(cid)))
        rfc.__dict__["_c_form"] = (fn, n, arr, rfc.tpb, rfc.blks, keep)
        return int(cid.value)

    def finish_and_sync(self) -> None:
        _chk(_lib.bodahip_finish_and_sync(self._ctx))

    def release_per_call_id_data(self) -> None:
        _chk(_lib.bodahip_release_per_call_id_data(self._ctx))

    def get_dur(self, b: int, e: int) -> float:
        ms = C.c_float()
        _chk(_lib.bodahip_get_dur(self._ctx, b, e, C.byref(ms)))
        return float(ms.value)

    def profile_start(self) -> None:
        _chk(_lib.bodahip_profile_start(self._ctx))

    def profile_stop(self) -> None:
        _chk(_lib.bodahip_profile_stop(self._ctx))

    def copy_nda_to_var(self, vn: str, nda: np.ndarray, dims: Optional[Dims] = None) -> None:
        d = dims or self.get_var_dims(vn)
        a = np.ascontiguousarray(nda, dtype=_NP[d.tn])
        if a.size != d.dims_prod():
            raise RtErr(f"copy_nda_to_var: {a.size} elements != dims {d.pretty()}")
        cd, ka = _cdims(d)
        _chk(_lib.bodahip_copy_to_var(self._ctx, vn.encode(), C.byref(cd), a.ctypes.data))
        self.finish_and_sync()  # the numpy buffer may die after return; the reference's host ndas are caller-owned

    def copy_var_to_nda(self, vn: str, dims: Optional[Dims] = None) -> np.ndarray:
        d = dims or self.get_var_dims(vn)
        out = np.empty(d.sizes if d.sizes else (1,), dtype=_NP[d.tn])
        cd, ka = _cdims(d)
        _chk(_lib.bodahip_copy_from_var(self._ctx, out.ctypes.data, C.byref(cd), vn.encode()))
        return out

    def get_var_raw_native_pointer(self, vn: str) -> int:
        p = C.c_void_p()
        _chk(_lib.bodahip_get_raw_ptr(self._ctx, vn.encode(), C.byref(p)))
        return int(p.value or 0)

    # ---- non-virtual conveniences (src/rtc_compute.cc:43-97)
    def create_var_from_nda(self, nda: np.ndarray, vn: str, dims: Dims) -> None:
        self.create_var_with_dims(vn, dims); self.copy_nda_to_var(vn, nda, dims)

    def create_nda_from_var(self, vn: str) -> np.ndarray:
        return self.copy_var_to_nda(vn)

    def init_var_from_vect_float(self, vn: str, v: np.ndarray) -> None:
        self.create_var_from_nda(np.asarray(v, np.float32), vn, Dims(("v",), (int(np.asarray(v).size),), "float"))

    # ---- additions (plumbing)
    def get_stream(self) -> int:
        p = C.c_void_p(); _chk(_lib.bodahip_get_stream(self._ctx, C.byref(p))); return int(p.value or 0)

    def get_device_info(self) -> dict:
        arch = C.create_string_buffer(64); cus = C.c_int(); khz = C.c_int()
        _chk(_lib.bodahip_get_device_info(self._ctx, arch, 64, C.byref(cus), C.byref(khz)))
        return {"arch": arch.value.decode(), "num_cus": cus.value, "clock_khz": khz.value}

    def set_tune(self, key: str, value: str) -> None:
        _chk(_lib.bodahip_set_tune(self._ctx, key.encode(), (value or "").encode()))

    # -- hipGraph capture of a call list (include/bodahip.h: bodahip_graph_*)
    def graph_begin(self) -> None:
        _chk(_lib.bodahip_graph_begin(self._ctx))

    def graph_end(self) -> Tuple[int, int]:
        """-> (graph id, number of captured calls)"""
        gid = C.c_uint32(); n = C.c_uint32()
        _chk(_lib.bodahip_graph_end(self._ctx, C.byref(gid), C.byref(n)))
        return int(gid.value), int(n.value)

    def graph_launch(self, graph_id: int) -> int:
        cid = C.c_uint32()
        _chk(_lib.bodahip_graph_launch(self._ctx, graph_id, C.byref(cid)))
        return int(cid.value)

    def graph_end_deps(self, deps: Sequence[Sequence[int]]) -> int:
        """End the capture with the true dependencies of its calls (deps[i] = earlier calls that call i must run after)."""
        ptr = [0]; idx: List[int] = []
        for d in deps:
            idx.extend(int(x) for x in d); ptr.append(len(idx))
        P = (C.c_uint32 * len(ptr))(*ptr); I = (C.c_uint32 * max(1, len(idx)))(*idx)
        gid = C.c_uint32()
        _chk(_lib.bodahip_graph_end_deps(self._ctx, len(deps), P, I, C.byref(gid)))
        return int(gid.value)

    def graph_destroy(self, graph_id: int) -> None:
        _chk(_lib.bodahip_graph_destroy(self._ctx, graph_id))

    def last_launch(self) -> dict:
        k = C.create_string_buffer(128); c = C.create_string_buffer(128); g = C.c_uint32(); b = C.c_uint32(); fl = C.c_double(); by = C.c_double()
        _chk(_lib.bodahip_last_launch(self._ctx, k, 128, c, 128, C.byref(g), C.byref(b), C.byref(fl), C.byref(by)))
        return {"kernel": k.value.decode(), "cfg": c.value.decode(), "grid": g.value, "block": b.value, "flops": fl.value, "algo_bytes": by.value}

    def torch_view(self, vn: str):
        """Zero-copy torch tensor over a var's device memory (for torch.distributed / RCCL collectives on weights).  bf16 storage is viewed as bytes (last dim x 2):
        the collectives move bytes, and gloo takes neither uint16 nor int16."""
        import torch
        d = self.get_var_dims(vn)
        ptr = self.get_var_raw_native_pointer(vn)
        tstr = {"float": "<f4", "half": "<f2", "bfloat16": "<i2", "int32_t": "<i4", "uint32_t": "<u4", "uint8_t": "|u1"}[d.tn]

        class _Holder:
            __cuda_array_interface__ = {"shape": tuple(d.sizes), "typestr": tstr, "data": (ptr, False), "version": 3, "strides": None}
        t = torch.as_tensor(_Holder(), device=f"cuda:{self.device_ordinal}")
        return t.view(torch.uint8) if d.tn == "bfloat16" else t


def compile_offline(src_or_opts: str, native_template: Optional[str] = None, arch: str = "gfx950", add_prelude: bool = True,
                    use_cache: bool = False) -> int:
    """hiprtc compile without a device (host-logic check; also used to pre-warm the code-object cache). -> code size."""
    sz = C.c_size_t(); log = C.create_string_buffer(1 << 16)
    rc = _lib.bodahip_compile_offline(src_or_opts.encode(), (native_template or "").encode() or None, arch.encode(),
                                      1 if add_prelude else 0, 1 if use_cache else 0, C.byref(sz), log, 1 << 16)
    _chk(rc)
    return int(sz.value)


def compile_to_file(src: str, out_path: str, arch: str = "gfx950", add_prelude: bool = True) -> None:
    """hiprtc-compile CUCL-dialect source to a code object file, without a device."""
    _chk(_lib.bodahip_compile_to_file(src.encode(), arch.encode(), 1 if add_prelude else 0, out_path.encode()))


def explain_plan(op: Op, num_cus: int = 256, tile: str = "") -> str:
    """The native planner's choice for an annotated op: '<kernel> <tile> <-D options>'.  Host-only, nothing is compiled."""
    buf = C.create_string_buffer(1 << 14)
    _chk(_lib.bodahip_explain_plan(op.to_str().encode(), num_cus, tile.encode(), buf, 1 << 14))
    return buf.value.decode()


def compile_stats() -> dict:
    """What run-time compilation cost this process so far: code objects served from the on-disk cache, compiled by hiprtc, and the time that took."""
    h = C.c_uint64(); m = C.c_uint64(); ms = C.c_double()
    _chk(_lib.bodahip_compile_stats(C.byref(h), C.byref(m), C.byref(ms)))
    return {"cache_hits": int(h.value), "compiled": int(m.value), "compile_ms": float(ms.value)}


def parse_op_native(line: str) -> str:
    """The C++ op-line parser of the backend (csrc/lexp.cc), canonical form back.  Host-only."""
    buf = C.create_string_buffer(1 << 16)
    _chk(_lib.bodahip_parse_op(line.encode(), buf, 1 << 16))
    return buf.value.decode()


def prebuild(op: Op, arch: str = "gfx950", num_cus: int = 256, tile: str = "") -> int:
    """AOT-compile into the code-object cache the specialisation run() would pick for `op`.  No GPU needed."""
    sz = C.c_size_t()
    _chk(_lib.bodahip_prebuild(op.to_str().encode(), arch.encode(), num_cus, tile.encode(), C.byref(sz)))
    return int(sz.value)


def make_rtc(spec: str = "(be=hip)", device_ordinal: int = 0) -> HipCompute:
    """NESI-style factory: '(be=hip)' -> backend (the reference creates backends from such lexps, src/rtc_prof.cc:204-217)."""
    from .op import parse_lexp
    kv = dict(parse_lexp(spec))
    if kv.get("be") not in ("hip", "cpu"):
        raise RtErr(f"unknown rtc back-end {kv.get('be')!r}; this package provides be=hip (and be=cpu, the host-cores baseline behind the same contract)")
    devs = None
    if "devices" in kv:   # "(be=hip,devices=0:1:2:3)" | "(be=hip,devices=all)": one backend over several GPUs (one host thread, one stream per GPU)
        if kv["be"] != "hip":
            raise RtErr("devices=... needs be=hip")
        if kv["devices"] == "all":
            import torch
            devs = list(range(max(1, torch.cuda.device_count())))
        else:
            devs = [int(x) for x in str(kv["devices"]).split(":")]
    return HipCompute(device_ordinal, kv["be"], devs)
