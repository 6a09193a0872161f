"""Variant selection for the hot path (host side; CPU only).

Restates the part of the reference's annotation layer that decides WHICH function runs an op and in what layout:
  * op_tune_t and its NESI text form                       src/cnn_op.H:14-30, dump rule src/nesi.cc:353-370
  * add_codegen_annotations / add_cnn_codegen_annotations  src/cnn_op.cc:16-68, 331-380
For be=hip every Convolution / sgemm is routed to the native side door -- `hip_conv` / `hip_sgemm` (or the reference's
own door names `cudnn_conv` / `cublas_sgemm` when use_culibs=1) -- which, exactly like use_culibs in the reference
(src/cnn_op.cc:47-48,142,339-340), leaves every tensor in reference layout and adds no `work` blocking.  The
reference's CUCL variants (conv/k1conv/tconv/ipconv/sgemm*) are produced by its own code generator and are reported
as unsupported here, the way ops-prof records any annotation failure (src/rtc_prof.cc:287-290).
"""
from __future__ import annotations
from dataclasses import dataclass, fields
from typing import Dict

from .op import Op, RtErr, UnsupErr, parse_lexp


@dataclass
class OpTune:
    use_be: str = ""
    use_culibs: int = 0
    MNt: tuple = (8, 8)
    MNb: tuple = (8, 16)
    Kb: int = 8
    use_local_mem: int = 1
    prof_variant: int = 0
    vw: int = 8
    k1conv: int = 0
    tconv: int = 0
    tconv_max_ksz: tuple = (11, 11)
    ipconv: int = 0
    hip_dtype: str = ""  # extension: "" / "f32" = exact fp32 MFMA path; "bf16" = bf16 operands, fp32 accumulate (BASELINE config 5)
    hip_algo: str = ""  # extension: "" = the bit-exact direct kernels; "winograd": func hip_conv_winograd (3x3 / stride-1 layers through F(2x2,3x3), mrd <= ~2e-3)
    hip_layout: str = ""  # extension (with hip_dtype=bf16): "nhwc" = channels-last bf16 STORAGE for in / filts / out: func hip_conv_nhwc on transposed operands,
    # the originals kept as <arg>_ref and filled / read back by xpose functions outside the timed call -- the reference's own k1conv / tconv protocol (boda_amd/nhwc.py)
    hip_s2d: int = 1  # extension (with hip_layout=nhwc): 1 = conv1-type layers (stride >= 2 on <= 8 channels) run space-to-depth, the regrouping done by the layout pass of `in` (boda_amd/nhwc.py)
    hip_patch: int = 1  # extension (with hip_layout=nhwc): 1 = layers with more than one tap and stride 1 in x take the LDS input-patch kernel (filts in the in_grp:y:x:out_chan:in_chan8 form); 0 = implicit GEMM for every layer
    hip_out: str = ""  # extension (with hip_layout=nhwc): "f32" = the kernel writes float instead of bfloat16
    hip_exact: int = 1  # extension: 1 = fp32 results bit-identical to the reference's per-thread fma chain (default); 0 = tolerance mode: within the reference's bound for re-associating
    # kernels (mrd < 2e-3, src/rtc_prof.cc:317-319,436; its 2e-4 default, :161, is not met by ANY second association of a K = 9216 sum on its U(-5,5) data) -- deterministic K slices on tile-starved long-K layers, Winograd where it is faster
    hip_tile: str = ""  # extension: workgroup tile of the native kernels "BIxBJxBKxWIxWJ[xMINW[xSPLITK[xMT[xPF[xSW[xKHO]]]]]]" ("" = heuristic)

    _ALWAYS = ("MNt", "MNb", "tconv_max_ksz")  # u32_pt_t fields print as "8 8" != default text "8:8": always dumped

    @staticmethod
    def parse(s: str) -> "OpTune":
        t = OpTune()
        items = parse_lexp(s) if s.strip() else []
        if isinstance(items, str):
            raise RtErr(f"op_tune: expected list, got {items!r}")
        names = {f.name for f in fields(OpTune)}
        for k, v in items:
            if k not in names or k.startswith("_"):
                raise RtErr(f"op_tune: unknown field {k!r}")
            if k in ("MNt", "MNb", "tconv_max_ksz"):
                parts = [p for p in str(v).replace(":", " ").split() if p]
                if len(parts) != 2:
                    raise RtErr(f"op_tune: {k} needs two values")
                setattr(t, k, (int(parts[0]), int(parts[1])))
            elif k in ("use_be", "hip_tile", "hip_dtype", "hip_algo", "hip_layout", "hip_out"):
                setattr(t, k, str(v))
            else:
                setattr(t, k, int(v))
        return t

    def to_str(self) -> str:
        """NESI dump: a field is printed unless equal to its default; u32_pt_t fields always (see module doc)."""
        d = OpTune()
        parts = []
        for f in fields(OpTune):
            if f.name.startswith("_"):
                continue
            v = getattr(self, f.name)
            if f.name in OpTune._ALWAYS:
                parts.append(f"{f.name}={v[0]} {v[1]}")
            elif v != getattr(d, f.name):
                parts.append(f"{f.name}={v}")
        return "(" + ",".join(parts) + ")"


REF_CUCL_VARIANTS = ("conv", "k1conv", "tconv", "ipconv", "conv_simd", "k1conv_simd")


def ref_conv_variant(op: Op, tune: OpTune) -> str:
    """Which CUCL variant the REFERENCE would pick for this conv under `tune` (src/cnn_op.cc:46-68); informational."""
    g = op.conv_geom()
    if tune.use_culibs:
        return "cudnn_conv"
    if tune.ipconv and g["PY"] == 0 and g["PX"] == 0 and g["OH"] == 1 and g["OW"] == 1:
        return "ipconv"
    if tune.k1conv and (g["KH"], g["KW"]) == (1, 1) and (g["SY"], g["SX"]) == (1, 1) and 6 <= g["OW"] <= 300 and g["OC"] >= 64:
        if g["PY"] or g["PX"]:
            return "conv"
        return "k1conv_simd" if tune.use_local_mem == 2 else "k1conv"
    if tune.tconv and (tune.tconv == 2 or (g["KW"] <= tune.tconv_max_ksz[0] and g["KH"] <= tune.tconv_max_ksz[1]
                                            and g["KW"] >= 1 and g["KH"] >= 1 and g["OW"] >= 6)):
        return "tconv"
    return "conv_simd" if tune.use_local_mem == 2 else "conv"


_TILE_WISDOM = None   # process-wide per-op best-tile table (boda_amd.wis_ana.TileWisdom), see set_tile_wisdom


def set_tile_wisdom(tw) -> None:
    """Install (or, with None, remove) the per-op best-tile table every later add_codegen_annotations consults: a TileWisdom, or the path of its text
    form (what `python -m boda_amd.wis_ana --tile-wisdom-out-fn` writes from the wisdom files ops-prof records).  The environment variable
    BODAHIP_TILE_WISDOM=<path> installs one at import."""
    global _TILE_WISDOM
    if isinstance(tw, str):
        from .wis_ana import TileWisdom
        tw = TileWisdom.load(tw)
    _TILE_WISDOM = tw


def add_codegen_annotations(op: Op, tune: OpTune, tile_wisdom=None) -> Op:
    """-> annotated copy of `op` with func_name (and conv_has_relu for convs) set.  Raises UnsupErr for variants
    this backend does not provide.  A tile recorded for this op in `tile_wisdom` (or the table installed with set_tile_wisdom) is given to
    the function when the tune names none: it then overrides the native planner's cost model for this function's calls only."""
    tw = tile_wisdom if tile_wisdom is not None else _TILE_WISDOM
    if tw is not None and not tune.hip_tile and not tune.hip_dtype and not tune.hip_layout and tune.hip_exact:   # (the table is recorded for the bit-exact fp32 functions)
        t = tw.tile_for(op)
        if t:
            import dataclasses
            tune = dataclasses.replace(tune, hip_tile=t)
    a = op.copy()
    t = a.get_type()
    native = (tune.use_be in ("", "hip"))
    if t == "Convolution":
        a.conv_geom()
        a.set_u32("conv_has_relu", 1)  # every Convolution under ops-prof (src/cnn_op.cc:337)
        if tune.use_culibs:
            a.set_func_name("cudnn_conv")
        elif native and not (tune.k1conv or tune.tconv or tune.ipconv) and tune.hip_dtype == "bf16" and tune.hip_layout == "nhwc":
            from . import nhwc
            nhwc.annotate(a, "float" if tune.hip_out == "f32" else "bfloat16", allow_s2d=bool(tune.hip_s2d), allow_patch=bool(tune.hip_patch))
        elif native and not (tune.k1conv or tune.tconv or tune.ipconv):
            a.set_func_name("hip_conv_bf16" if tune.hip_dtype == "bf16" else ("hip_conv_winograd" if tune.hip_algo == "winograd" else "hip_conv"))
        else:
            raise UnsupErr(f"variant '{ref_conv_variant(op, tune)}' is generated by the reference's CUCL code generator; "
                           f"be=hip provides hip_conv / cudnn_conv (op_tune={tune.to_str()})")
    elif t == "sgemm":
        a.sgemm_geom()
        if tune.use_culibs:
            a.set_func_name("cublas_sgemm")
        elif native:
            a.set_func_name("hip_sgemm_bf16" if tune.hip_dtype == "bf16" else "hip_sgemm")
        else:
            raise UnsupErr(f"sgemm variants of use_be={tune.use_be!r} are generated by the reference's CUCL code generator")
    else:
        raise UnsupErr(f"op type {t!r} is not on the conv_fwd / sgemm hot path")
    if not tune.hip_exact and native and not tune.use_culibs and tune.hip_dtype != "bf16":
        a.str_vals["hip_exact"] = "0"   # travels with the function, like hip_tile
    if tune.hip_tile and native and not tune.use_culibs:
        a.str_vals["hip_tile"] = tune.hip_tile   # travels with the function: the backend applies it to this function's calls only
    return a


# arg tables of the native side-door functions (the stubs test/rtc/cublas_sgemm.cucl:1-4, cudnn_conv.cucl:1-7)
NATIVE_ARGS: Dict[str, tuple] = {
    "hip_sgemm": (("a", "IN"), ("b", "IN"), ("c", "OUT")),
    "cublas_sgemm": (("a", "IN"), ("b", "IN"), ("c", "OUT")),
    "hip_sgemm_bf16": (("a", "IN"), ("b", "IN"), ("c", "OUT")),
    "hip_conv_bf16": (("filts", "IN"), ("biases", "IN"), ("in", "IN"), ("stride", "REF"), ("in_pad", "REF"), ("out", "OUT")),
    "hip_conv": (("filts", "IN"), ("biases", "IN"), ("in", "IN"), ("stride", "REF"), ("in_pad", "REF"), ("out", "OUT")),
    "cudnn_conv": (("filts", "IN"), ("biases", "IN"), ("in", "IN"), ("stride", "REF"), ("in_pad", "REF"), ("out", "OUT")),
    "hip_conv_winograd": (("filts", "IN"), ("biases", "IN"), ("in", "IN"), ("stride", "REF"), ("in_pad", "REF"), ("out", "OUT")),
    "hip_conv_nhwc": (("filts", "IN"), ("biases", "IN"), ("in", "IN"), ("stride", "REF"), ("in_pad", "REF"), ("out", "OUT")),
    "hip_conv_k1_chain": (("filts", "IN"), ("biases", "IN"), ("filts2", "IN"), ("biases2", "IN"), ("in", "IN"), ("stride", "REF"), ("in_pad", "REF"), ("out", "OUT")),
    "hip_conv_filts_kmajor": (("filts", "IN"), ("filts_km", "OUT")),   # filts as [K + 128][out_chan padded to 4]: what hip_conv's optional filts_km arg takes
}


K1_CHAIN_FUNC = "hip_conv_k1_chain"
FILTS_KMAJOR_FUNC = "hip_conv_filts_kmajor"


def k1_chain_applies(a: Op, b: Op) -> bool:
    """Can the fp32 convolutions a -> b (b reads a's output) run as one hip_conv_k1_chain launch?  Mirrors plan_k1_chain (csrc/native_kernels.cc): both 1x1 /
    stride 1 / unpadded, at most 96 intermediate channels, at most 128 out_chans, both filter images (k-major, odd pitch) within the LDS."""
    ga, gb = a.conv_geom(), b.conv_geom()
    for g in (ga, gb):
        if (g["KH"], g["KW"], g["SY"], g["SX"], g["PY"], g["PX"]) != (1, 1, 1, 1, 0, 0):
            return False
    if gb["C"] != ga["OC"] or (gb["H"], gb["W"], gb["B"]) != (ga["OH"], ga["OW"], ga["B"]) or ga["OH"] * ga["OW"] < 4:
        return False
    if not (1 <= ga["OC"] <= 96 and 1 <= gb["OC"] <= 128):
        return False
    ocb, ocb2 = -(-ga["OC"] // 32), -(-gb["OC"] // 32)
    kp, kp2 = (ga["C"] + 1) // 2 * 2, (ga["OC"] + 1) // 2 * 2
    return 4 * (kp * ((ocb * 32) | 1) + ocb * 32 + kp2 * ((ocb2 * 32) | 1) + ocb2 * 32) <= 160 * 1024


def annotate_k1_chain(a: Op, b: Op, relu_a: int, relu_b: int) -> Op:
    """The function op of hip_conv_k1_chain for the annotated hip_conv ops a -> b: a's in / filts / biases / stride / in_pad, b's filts / biases as filts2 /
    biases2, b's out; conv_has_relu / conv_has_relu2 the two fused ReLUs."""
    if not k1_chain_applies(a, b):
        raise UnsupErr("hip_conv_k1_chain: two chained 1x1 / stride-1 / unpadded convolutions with <= 96 intermediate channels and <= 128 out_chans")
    c = a.copy()
    c.nda_vals["filts2"] = b.nda_vals["filts"]; c.nda_vals["biases2"] = b.nda_vals["biases"]; c.nda_vals["out"] = b.nda_vals["out"]
    if "out_chans" in b.nda_vals:
        c.nda_vals["out_chans"] = b.nda_vals["out_chans"]
    from .op import Nda
    c.nda_vals["conv_has_relu"] = Nda(None, "uint32_t", (int(relu_a),)); c.nda_vals["conv_has_relu2"] = Nda(None, "uint32_t", (int(relu_b),))
    c.str_vals["func_name"] = K1_CHAIN_FUNC
    c.str_vals.pop("hip_tile", None)
    return c


def f32_pool_fusable(conv: Op, pool_in, kern, stride, in_pad, avg_pool) -> bool:
    """Can a max pooling (input dims pool_in, window kern, stride, padding in_pad) be taken into the annotated fp32 hip_conv function that alone reads its output?
    Mirrors apply_f32_pool (csrc/native_run.cc) / plan_conv (csrc/native_plan.cc): max, no pooling pad, windows of at most 3 x 3 that tile the plane exactly (no clipped last window),
    and a convolution that takes the LDS-patch form (stride 1 in x, more than one tap, not output 1 x 1)."""
    if avg_pool or tuple(in_pad) != (0, 0) or conv.get_func_name() != "hip_conv" or "hip_tile" in conv.str_vals or conv.has("hip_pool"):
        return False
    kh, kw = kern; sy, sx = stride; H, W = pool_in.dsz("y"), pool_in.dsz("x")
    if not (1 <= kh <= 3 and 1 <= kw <= 3 and kh * kw >= 2 and sy >= 1 and sx >= 1 and H >= kh and W >= kw and (H - kh) % sy == 0 and (W - kw) % sx == 0):
        return False
    g = conv.conv_geom()
    if (g["H"], g["W"]) != ((H - kh) // sy + 1, (W - kw) // sx + 1) or pool_in.dsz("chan") != g["C"] or pool_in.dsz("img") != g["B"]:
        return False
    return g["SX"] == 1 and g["KH"] * g["KW"] >= 2 and g["KH"] >= g["SY"] and not (g["OH"] == 1 and g["OW"] == 1)


def fuse_f32_pool(conv: Op, pool_in, kern, stride) -> None:
    """In place: the annotated fp32 hip_conv function takes the max pooling in front of it (f32_pool_fusable).  Its `in` becomes the POOLING's input and the window travels
    with the function: uint32 hip_pool, dims pool_sz / pool_stride.  The reference runs two functions (test/rtc/pool.cucl, then the conv: src/rtc_fwd.cc:545-549); here a
    patch element of the convolution's LDS input patch is the window maximum, formed while the patch is staged (kernels/gemm_conv_f32.hip, PKH).  Bit-identical."""
    from .op import Dims, Nda
    none = lambda y, x: Nda(Dims(("y", "x"), (y, x), "none"), "none")
    conv.nda_vals["in"] = Nda(dims=pool_in, tn=pool_in.tn)
    conv.set_u32("hip_pool", 1)
    conv.nda_vals["pool_sz"] = none(*kern); conv.nda_vals["pool_stride"] = none(*stride)


import os as _os
if _os.environ.get("BODAHIP_TILE_WISDOM"):
    set_tile_wisdom(_os.environ["BODAHIP_TILE_WISDOM"])
