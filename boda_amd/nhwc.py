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
import os
from typing import Dict, List, Tuple

from .op import Dims, Nda, Op, UnsupErr
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


def s2d_geom(g: dict):
    """Space-to-depth form of a conv1-type layer (stride s in both axes on <= 8 channels, kernel wider than the stride: 11x11 / 4, 7x7 / 2 on
    3 channels): an s x s block of input pixels becomes s*s channels,
        in2[b][Y][X][c*s*s + dy*s + dx]   = in[b][c][s*Y + dy - Pry][s*X + dx - Prx]                    (zero outside; Pr = pad rounded up to a multiple of s)
        f2 [oc][a][b][c*s*s + dy*s + dx]  = filts[oc][c][s*a + dy - (Pry - PY)][s*b + dx - (Prx - PX)]  (zero outside)
    and the layer is the stride-1, unpadded KH2 x KW2 convolution of in2 with f2 -- the same sums term for term, with C*s*s contiguous
    channels per position instead of C (8 stored) and ceil-ed kernel extents.  -> dict or None."""
    s = g["SY"]
    if not (2 <= s <= 4 and g["SX"] == s and g["KH"] > s and g["KW"] > s and g["C"] <= 8 and g["C"] * s * s <= 64):
        return None
    pry, prx = (g["PY"] + s - 1) // s * s, (g["PX"] + s - 1) // s * s
    kh2, kw2 = (g["KH"] + (pry - g["PY"]) + s - 1) // s, (g["KW"] + (prx - g["PX"]) + s - 1) // s
    return dict(S=s, PRY=pry, PRX=prx, KH2=kh2, KW2=kw2, C2=g["C"] * s * s, H2=g["OH"] + kh2 - 1, W2=g["OW"] + kw2 - 1)


PATCH_LDS_LIMIT = 160 * 1024
_PATCH_TILES = ((128, 128), (64, 256), (64, 128), (128, 64), (32, 128), (128, 256), (256, 128))   # (BI, BJ) of plan_conv_nhwc_patch's candidates


def patch_min_lds(KH: int, SY: int, W: int, PX: int, OH: int, OW: int, out_f32: bool) -> int:
    """Bytes of LDS the input-patch kernel needs at least for this plane: plan_conv_nhwc_patch's own bound (csrc/native_plan.cc, `lds_cg`: the double-buffered
    patch of ONE channel group -- the zero-padded input rows a tile's output positions touch, at the kernel's slot pitch -- or the epilogue's transposed tile),
    minimised over the planner's tiles.  The layout of `filts` is chosen when the op is annotated and binds the kernel at run time, so the annotation must know
    whether any tile fits before it asks for the F' form."""
    wp = W + 2 * PX
    for p2 in range(wp, wp + 16):
        if (SY * p2 - OW) % 16 == 0:
            wp = p2; break
    best = None
    for bi, bj in _PATCH_TILES:
        rows_max = (bj - 2) // OW + 2
        seg_max = min((OH - 1 + rows_max - 1) // OH + 1, rows_max)
        cs = ((rows_max - seg_max) * SY + seg_max * KH) * wp
        csp = cs + ((2 - cs % 16) + 16) % 16
        need = max(2 * 16 * csp, 0 if out_f32 else bj * (bi * 2 + 16))
        best = need if best is None else min(best, need)
    return best


def patch_eligible(g: dict, out_f32: bool = False) -> bool:
    """Layers that run from an LDS input patch (kernels/conv_nhwc_patch_bf16.hip): more than one tap, stride 1 in x, windows that overlap or abut in y, an output
    map (not the whole-input kernels of fully-connected layers, which stay implicit GEMMs with K slices) -- and a plane narrow enough for the patch of one channel
    group to fit the LDS (5x5 on ~500 columns, 3x3 on ~850 do not: those stay on the implicit GEMM, which has no such bound)."""
    return (g["KH"] * g["KW"] >= 2 and g["SX"] == 1 and g["KH"] >= g["SY"] and g["OH"] * g["OW"] > 1 and
            patch_min_lds(g["KH"], g["SY"], g["W"], g["PX"], g["OH"], g["OW"], out_f32) <= PATCH_LDS_LIMIT)


def patch_filts_dims(f: Dims) -> Dims:
    """out_chan:in_chan:y:x -> in_grp:y:x:out_chan:in_chan8: F'[g][ky][kx][oc][8] = filts[oc][8g .. 8g+8)[ky][kx] (zero past in_chan)."""
    return Dims(("in_grp", "y", "x", "out_chan", "in_chan8"), (pad8(f.dsz("in_chan")) // 8, f.dsz("y"), f.dsz("x"), f.dsz("out_chan"), 8), "bfloat16")


def annotate(a: Op, out_tn: str = "bfloat16", allow_s2d: bool = True, allow_patch: bool = True) -> None:
    """In place: the `hip_conv_nhwc` form of an annotated Convolution -- kernel dims for in / filts / out, the originals as <arg>_ref.
    conv1-type layers additionally go space-to-depth (s2d_geom): the kernel then sees a stride-1, unpadded convolution; the original
    stride / in_pad / kern_sz stay as <arg>_ref and the scalars nhwc_s2d{,_pry,_prx} tell the layout passes how `in` / `filts` are filled."""
    g = a.conv_geom()
    sd = s2d_geom(g) if allow_s2d else None
    for an in ("in", "filts", "out"):
        ref = a.get_dims(an)
        a.nda_vals[an + "_ref"] = Nda(dims=ref, tn=ref.tn)
    i, f, o = a.get_dims("in_ref"), a.get_dims("filts_ref"), a.get_dims("out_ref")
    if sd is None:
        a.nda_vals["in"] = Nda(dims=nhwc_dims(i), tn="bfloat16")
        # the filters' layout selects the kernel: F' for the input-patch kernel (3x3 / 5x5 ... stride-1-in-x layers), out_chan:y:x:in_chan for the implicit GEMM
        a.nda_vals["filts"] = Nda(dims=patch_filts_dims(f) if (allow_patch and patch_eligible(g, out_tn == "float")) else ohwi_dims(f), tn="bfloat16")
    else:
        c2p = pad8(sd["C2"])
        # (a plane too wide for the LDS patch keeps the space-to-depth form but runs it on the implicit GEMM)
        allow_patch = allow_patch and patch_min_lds(sd["KH2"], 1, sd["W2"], 0, g["OH"], g["OW"], out_tn == "float") <= PATCH_LDS_LIMIT
        a.nda_vals["in"] = Nda(dims=Dims(("img", "y", "x", "chan"), (g["B"], sd["H2"], sd["W2"], c2p), "bfloat16"), tn="bfloat16")
        # (the space-to-depth form is a stride-1 KH2 x KW2 convolution: the input-patch kernel's case)
        a.nda_vals["filts"] = Nda(dims=Dims(("in_grp", "y", "x", "out_chan", "in_chan8"), (c2p // 8, sd["KH2"], sd["KW2"], g["OC"], 8), "bfloat16") if allow_patch else
                                  Dims(("out_chan", "y", "x", "in_chan"), (g["OC"], sd["KH2"], sd["KW2"], c2p), "bfloat16"), tn="bfloat16")
        none = lambda y, x: Nda(Dims(("y", "x"), (y, x), "none"), "none")
        for an, v in (("stride", none(1, 1)), ("in_pad", none(0, 0)), ("kern_sz", none(sd["KH2"], sd["KW2"]))):
            if a.has(an):
                a.nda_vals[an + "_ref"] = a.nda_vals[an]
            a.nda_vals[an] = v
        a.set_u32("nhwc_s2d", sd["S"]); a.set_u32("nhwc_s2d_pry", sd["PRY"]); a.set_u32("nhwc_s2d_prx", sd["PRX"])
    a.nda_vals["out"] = Nda(dims=nhwc_dims(o, out_tn, pad=False), tn=out_tn)
    a.set_func_name(FUNC)


def pool_fusable(g: dict, in_hw: Tuple[int, int], kern: Tuple[int, int], stride: Tuple[int, int], pad: Tuple[int, int], avg: bool, out_f32: bool = False) -> bool:
    """Can a pooling (window kern, stride, pad; on planes in_hw) be taken into the 1x1 convolution `g` that consumes it?  Max pooling with stride 1 in front of a
    1x1 / stride-1 / unpadded convolution whose input plane is the pooling's output, a window of 2..25 positions, and a plane narrow enough for the LDS patch.
    (The caller also owes non-negative pooling input: see kernels/conv_nhwc_patch_bf16.hip, POOL.)"""
    if avg or tuple(stride) != (1, 1) or not (2 <= kern[0] * kern[1] <= 25):
        return False
    if not (g["KH"] == g["KW"] == 1 and g["SY"] == g["SX"] == 1 and g["PY"] == g["PX"] == 0):
        return False
    H, W = in_hw
    if (H + 2 * pad[0] - kern[0] + 1, W + 2 * pad[1] - kern[1] + 1) != (g["H"], g["W"]) or g["OH"] * g["OW"] <= 1:
        return False
    return patch_min_lds(kern[0], 1, W, pad[1], g["OH"], g["OW"], out_f32) <= PATCH_LDS_LIMIT


def fuse_pool(a: Op, pool_in: Dims, kern: Tuple[int, int], pad: Tuple[int, int]) -> None:
    """In place: an annotated 1x1 hip_conv_nhwc function takes the max pooling in front of it (pool_fusable).  Its `in` becomes the POOLING's input, its filters take
    the input-patch form (one k-slot per channel group), and the window travels with the function: uint32 nhwc_pool, dims pool_sz / pool_pad.  The reference
    runs the two as two functions (test/rtc/pool.cucl, then the conv; src/rtc_fwd.cc:545-549); here the window maximum is taken while the MFMA B fragment is formed."""
    if a.get_func_name() != FUNC or a.has("nhwc_s2d") or a.get_dims("filts").has("in_grp"):
        raise UnsupErr("fuse_pool: needs a plain hip_conv_nhwc function")
    none = lambda y, x: Nda(Dims(("y", "x"), (y, x), "none"), "none")
    a.nda_vals["filts"] = Nda(dims=patch_filts_dims(a.get_dims("filts_ref")), tn="bfloat16")
    a.nda_vals["in_ref"] = Nda(dims=pool_in, tn=pool_in.tn)
    a.nda_vals["in"] = Nda(dims=nhwc_dims(pool_in), tn="bfloat16")
    a.set_u32("nhwc_pool", 1)
    a.nda_vals["pool_sz"] = none(*kern); a.nda_vals["pool_pad"] = none(*pad)


def post_fusable(a: Op, kern, stride, pad, avg: bool, lrn=None) -> bool:
    """Can the max pooling that alone reads this convolution's output -- and the across-channel LRN that alone reads the pooling's (lrn = (local_size, alpha, beta, k)) -- run
    INSIDE the convolution's launch?  The rolling-rows kernel's conditions (csrc/kernels/conv_nhwc_rows_bf16.hip; the planner has the last word: explain_plan): the LDS-patch
    form of filts, stride 1 (after space-to-depth), at most 64 out_chans in whole 8-channel chunks, a bfloat16 output, a fused ReLU (the caller checks), no pooling in front."""
    if a.get_func_name() != FUNC or not a.get_dims("filts").has("in_grp") or a.has("nhwc_pool") or a.has("nhwc_post_pool") or a.get_dims("out").tn != "bfloat16":
        return False
    st, oc = a.get_dims("stride"), a.get_dims("filts").dsz("out_chan")
    if (st.dsz("y"), st.dsz("x")) != (1, 1) or oc > 64 or oc % 8:
        return False
    if avg or not (1 <= kern[0] <= 7 and 1 <= kern[1] <= 7 and kern[0] * kern[1] >= 2) or min(stride) < 1 or not (0 <= pad[0] < kern[0] and 0 <= pad[1] < kern[1]):
        return False
    if lrn is not None:
        ls, alpha, _, k = lrn
        if not (ls % 2 == 1 and 1 <= ls // 2 <= 4 and k > 0.0 and alpha >= 0.0):
            return False
    return True


def fuse_post(a: Op, pool_out: Dims, kern, stride, pad, lrn=None) -> None:
    """In place: the annotated hip_conv_nhwc function takes the max pooling behind it (post_fusable) and, with lrn = (local_size, alpha, beta, k), the LRN behind that.  Its
    `out` becomes the POOLED tensor; window and LRN constants travel with the function (uint32 nhwc_post_pool, dims post_pool_sz / post_pool_stride / post_pool_pad, uint32
    nhwc_post_lrn = local size, floats post_lrn_alpha / post_lrn_beta / post_lrn_k).  The reference runs conv, pool and lrn as three functions (src/rtc_fwd.cc:495-503,
    545-549, test/rtc/pool.cucl, lrn.cucl); here the convolution's rows are pooled out of an LDS ring and its own output tensor is never written."""
    none = lambda y, x: Nda(Dims(("y", "x"), (y, x), "none"), "none")
    a.nda_vals["out_ref"] = Nda(dims=pool_out, tn=pool_out.tn)
    a.nda_vals["out"] = Nda(dims=nhwc_dims(pool_out, "bfloat16", pad=False), tn="bfloat16")
    a.set_u32("nhwc_post_pool", 1)
    a.nda_vals["post_pool_sz"] = none(*kern); a.nda_vals["post_pool_stride"] = none(*stride); a.nda_vals["post_pool_pad"] = none(*pad)
    if lrn is not None:
        ls, alpha, beta, k = lrn
        a.set_u32("nhwc_post_lrn", int(ls))
        for n, v in (("post_lrn_alpha", alpha), ("post_lrn_beta", beta), ("post_lrn_k", k)):
            a.nda_vals[n] = Nda(None, "float", (float(v),))


GRP_FUNC = "hip_conv_nhwc_grp"


def group_pad(ocs) -> int:
    """Padding granularity of the members of a horizontally fused convolution: the largest of 128 / 64 / 32 out_chans that adds at most a quarter of zero
    rows (a tile row must belong to one member, so tiles are no taller than this)."""
    for gp in (128, 64, 32):
        if sum(-(-o // gp) * gp for o in ocs) <= 1.25 * sum(ocs):
            return gp
    return 32


def annotate_group(annos: List[Op]) -> Op:
    """`hip_conv_nhwc_grp`: up to four annotated hip_conv_nhwc ops that read the same `in` with the same kernel geometry, as ONE function -- filts / biases
    stacked along out_chan (member m at rows [oc0_m, oc0_m + out_chans_m), oc0_m = the earlier members' out_chans each rounded up to grp.pad), outputs
    out_0 .. out_{n-1}.  The horizontal counterpart of what the reference does vertically when it fuses a ReLU into its conv (src/rtc_fwd.cc:486-493)."""
    a0 = annos[0]
    if not (2 <= len(annos) <= 4):
        raise UnsupErr("hip_conv_nhwc_grp: 2..4 members")
    for a in annos:
        if a.get_func_name() != FUNC or a.has("nhwc_s2d") or a.get_dims("filts").has("in_grp"):
            raise UnsupErr("hip_conv_nhwc_grp: members must be plain hip_conv_nhwc functions")
        for an in ("in", "stride", "in_pad", "kern_sz"):
            if a.get_dims(an) != a0.get_dims(an):
                raise UnsupErr(f"hip_conv_nhwc_grp: members differ in {an}")
        if a.get_u32("conv_has_relu") != a0.get_u32("conv_has_relu") or a.get_dims("out").tn != a0.get_dims("out").tn:
            raise UnsupErr("hip_conv_nhwc_grp: members differ in ReLU / output type")
    ocs = [a.get_dims("filts").dsz("out_chan") for a in annos]
    gp = group_pad(ocs); tot = sum(-(-o // gp) * gp for o in ocs)
    f0 = a0.get_dims("filts")
    nv = {"in": a0.nda_vals["in"], "stride": a0.nda_vals["stride"], "in_pad": a0.nda_vals["in_pad"], "kern_sz": a0.nda_vals["kern_sz"],
          "filts": Nda(dims=Dims(f0.names, (tot,) + tuple(f0.sizes[1:]), "bfloat16"), tn="bfloat16"), "biases": Nda(dims=Dims(("out_chan",), (tot,), "float"), tn="float"),
          "grp": Nda(dims=Dims(tuple(f"m{m}" for m in range(len(ocs))) + ("pad",), tuple(ocs) + (gp,), "none"), tn="none"),
          "conv_has_relu": a0.nda_vals["conv_has_relu"]}
    for m, a in enumerate(annos):
        nv[f"out_{m}"] = a.nda_vals["out"]
    return Op({"type": "Convolution", "func_name": GRP_FUNC}, nv)


MULTI_FUNC = "hip_conv_nhwc_multi"
_MULTI_MEMBER_ARGS = ("filts", "biases", "in", "stride", "in_pad", "out")


def multi_eligible(anno: Op) -> bool:
    """A member of a multi-problem launch: a plain hip_conv_nhwc function on the implicit-GEMM kernel (filts out_chan:y:x:in_chan) -- not the input-patch form, not
    space-to-depth (both bind other kernels with another summation order)."""
    return anno.get_func_name() == FUNC and not anno.has("nhwc_s2d") and not anno.get_dims("filts").has("in_grp") and not anno.has("nhwc_pool") and not anno.has("nhwc_post_pool")


def annotate_multi(annos: List[Op]) -> Op:
    """`hip_conv_nhwc_multi`: up to 256 annotated hip_conv_nhwc ops -- INDEPENDENT convolutions, each with its own tensors and its own geometry -- as ONE function whose
    launch hands all their tiles to the chip together (kernels/conv_nhwc_multi_bf16.hip).  Member m's args carry the suffix _<m>.  Per member the arithmetic is that of
    its own hip_conv_nhwc call on the implicit-GEMM kernel, bit for bit.  The reference runs one function per op (src/rtc_fwd.cc:545-549); this is the launch-bound
    regime's answer on MI355X (54-64 launches per config-5 list, most of them 50-200 tiles for 256 CUs)."""
    if not (1 <= len(annos) <= 256):
        raise UnsupErr("hip_conv_nhwc_multi: 1..256 members")
    nv = {"multi": Nda(dims=Dims(("n",), (len(annos),), "none"), tn="none")}
    relu = [a.get_u32("conv_has_relu") for a in annos]
    for m, a in enumerate(annos):
        if not multi_eligible(a):
            raise UnsupErr("hip_conv_nhwc_multi: members must be plain hip_conv_nhwc functions (out_chan:y:x:in_chan filters)")
        if a.get_dims("out").tn != annos[0].get_dims("out").tn:
            raise UnsupErr("hip_conv_nhwc_multi: members differ in output type")
        for an in _MULTI_MEMBER_ARGS + ("kern_sz",):
            nv[f"{an}_{m}"] = a.nda_vals[an]
    nv["conv_has_relu"] = Nda(dims=None, tn="uint32_t", v=(int(all(relu)),))
    if any(relu) != all(relu):
        if len(annos) > 32:
            raise UnsupErr("hip_conv_nhwc_multi: more than 32 members must agree in ReLU")
        nv["relu_mask"] = Nda(dims=None, tn="uint32_t", v=(sum(int(bool(r)) << m for m, r in enumerate(relu)),))
    return Op({"type": "Convolution", "func_name": MULTI_FUNC}, nv)


SET_FUNC = "hip_conv_nhwc_set"


def set_eligible(anno: Op) -> bool:
    """A member of a set: any hip_conv_nhwc function -- implicit-GEMM or input-patch form of filts (the member keeps its own specialised kernel code); not the
    space-to-depth conv1 form, whose input layout belongs to the net's first layout pass."""
    return anno.get_func_name() == FUNC and not anno.has("nhwc_s2d") and not anno.has("nhwc_post_pool")     # (nor a convolution with a pooling taken into its launch: the rolling-rows kernel)


def annotate_set(annos: List[Op]) -> Op:
    """`hip_conv_nhwc_set`: 2..16 annotated hip_conv_nhwc ops -- independent convolutions -- as ONE function whose launch runs every member on ITS OWN specialised
    kernel code (kernels/conv_nhwc_bf16.hip or conv_nhwc_patch_bf16.hip, instantiated per member inside one wrapper kernel that the backend builds at run time).
    Where hip_conv_nhwc_multi trades the specialisation for any number of members, a set keeps it for a few: an inception module's 3x3 / 5x5 / pool-projection
    convolutions, which are three launches of 100-200 tiles each on 256 CUs.  Args as hip_conv_nhwc_multi (suffix _<m>); results are those of the members' own
    launches, bit for bit (members that would slice K on their own run unsliced here: the other members fill the chip)."""
    if not (2 <= len(annos) <= 16):
        raise UnsupErr("hip_conv_nhwc_set: 2..16 members")
    nv = {"multi": Nda(dims=Dims(("n",), (len(annos),), "none"), tn="none")}
    relu = [a.get_u32("conv_has_relu") for a in annos]
    out_tn = lambda a: a.get_dims("out_0" if a.get_func_name() == GRP_FUNC else "out").tn
    for m, a in enumerate(annos):
        if out_tn(a) != out_tn(annos[0]):
            raise UnsupErr("hip_conv_nhwc_set: members differ in output type")
        if a.get_func_name() == GRP_FUNC:     # a horizontally fused member (hip_conv_nhwc_grp): its args keep their names, plus the member's suffix
            for an, v in a.nda_vals.items():
                if an != "conv_has_relu":
                    nv[f"{an}_{m}"] = v
            continue
        if not set_eligible(a):
            raise UnsupErr("hip_conv_nhwc_set: members must be hip_conv_nhwc / hip_conv_nhwc_grp functions (not the space-to-depth form)")
        for an in _MULTI_MEMBER_ARGS + ("kern_sz",):
            nv[f"{an}_{m}"] = a.nda_vals[an]
        if a.has("nhwc_pool"):     # (a member with max pooling fused in front of it)
            for an in ("nhwc_pool", "pool_sz", "pool_pad"):
                nv[f"{an}_{m}"] = a.nda_vals[an]
    nv["conv_has_relu"] = Nda(dims=None, tn="uint32_t", v=(int(all(relu)),))
    if any(relu) != all(relu):     # (ReLU is a constant of each member's own kernel instantiation: members may differ)
        nv["relu_mask"] = Nda(dims=None, tn="uint32_t", v=(sum(int(bool(r)) << m for m, r in enumerate(relu)),))
    return Op({"type": "Convolution", "func_name": SET_FUNC}, nv)


def multi_arg_names(n: int) -> List[str]:
    return ["multi"] + [f"{an}_{m}" for m in range(n) for an in _MULTI_MEMBER_ARGS]


def group_arg_names(n: int) -> List[str]:
    return ["filts", "biases", "in", "stride", "in_pad", "grp"] + [f"out_{m}" for m in range(n)]


def group_row_offsets(grp: Dims) -> List[int]:
    gp = grp.dsz("pad"); offs, t = [], 0
    for m in range(len(grp.sizes) - 1):
        offs.append(t); t += -(-grp.sizes[m] // gp) * gp
    return offs


# layout passes (one thread per element of the destination; sizes by value).  `__bf16` conversions round to nearest even.
XPOSE_SRC = """
typedef __bf16 xp_bf16x8_t __attribute__((ext_vector_type(8)));
// in_ref img:chan:y:x float -> in img:y:x:chan bf16, one thread per 16-byte chunk (8 channels) of the destination.  With S > 1 the
// destination is the space-to-depth tensor: channel c2 = c*S*S + dy*S + dx of position (Y, X) is pixel (S*Y + dy - PRY, S*X + dx - PRX) of
// channel c (zero outside the plane); S = 1, PR = 0 is the plain transposition.  Channels >= C2 are zero pad.
CUCL_GLOBAL_KERNEL void hip_conv_nhwc_xpose_in( GASQ float const * const in_ref, GASQ xp_bf16x8_t * const in, uint32_t const n, uint32_t const C,
                                                 uint32_t const H, uint32_t const W, uint32_t const C2, uint32_t const C8, uint32_t const H2,
                                                 uint32_t const W2, uint32_t const S, uint32_t const PRY, uint32_t const PRX ) {
  // CUCL IX GLOB_ID_1D in n=n
  uint32_t const i = GLOB_ID_1D;
  if( i >= n ) { return; }
  uint32_t const q = i % C8, X = ( i / C8 ) % W2, Y = ( i / ( C8*W2 ) ) % H2, img = i / ( C8*W2*H2 );
  xp_bf16x8_t r;
  for( uint32_t e = 0; e != 8; ++e ) {
    uint32_t const c2 = 8*q + e, c = c2 / ( S*S ), dy = ( c2 / S ) % S, dx = c2 % S;
    int32_t const y = (int32_t)( S*Y + dy ) - (int32_t)PRY, x = (int32_t)( S*X + dx ) - (int32_t)PRX;
    bool const ok = ( c2 < C2 ) && ( y >= 0 ) && ( y < (int32_t)H ) && ( x >= 0 ) && ( x < (int32_t)W );
    r[e] = (__bf16)( ok ? in_ref[( ( img*C + c )*H + y )*W + x] : 0.0f );
  }
  in[i] = r;
}
// filts_ref out_chan:in_chan:y:x float -> filts out_chan:y:x:in_chan bf16 (S > 1: the space-to-depth filters; tap (a, b), channel c2 is tap
// (S*a + dy - OFY, S*b + dx - OFX) of channel c, zero outside the kernel)
CUCL_GLOBAL_KERNEL void hip_conv_nhwc_xpose_filts( GASQ float const * const filts_ref, GASQ xp_bf16x8_t * const filts, uint32_t const n, uint32_t const C,
                                                    uint32_t const KH, uint32_t const KW, uint32_t const C2, uint32_t const C8, uint32_t const KH2,
                                                    uint32_t const KW2, uint32_t const S, uint32_t const OFY, uint32_t const OFX ) {
  uint32_t const i = GLOB_ID_1D;
  if( i >= n ) { return; }
  uint32_t const q = i % C8, b = ( i / C8 ) % KW2, a = ( i / ( C8*KW2 ) ) % KH2, oc = i / ( C8*KW2*KH2 );
  xp_bf16x8_t r;
  for( uint32_t e = 0; e != 8; ++e ) {
    uint32_t const c2 = 8*q + e, c = c2 / ( S*S ), dy = ( c2 / S ) % S, dx = c2 % S;
    int32_t const y = (int32_t)( S*a + dy ) - (int32_t)OFY, x = (int32_t)( S*b + dx ) - (int32_t)OFX;
    bool const ok = ( c2 < C2 ) && ( y >= 0 ) && ( y < (int32_t)KH ) && ( x >= 0 ) && ( x < (int32_t)KW );
    r[e] = (__bf16)( ok ? filts_ref[( ( oc*C + c )*KH + y )*KW + x] : 0.0f );
  }
  filts[i] = r;
}
// filts_ref out_chan:in_chan:y:x float -> F' in_grp:y:x:out_chan:in_chan8 bf16 (the input-patch kernel's filters): one thread per 16-byte chunk (g, ky, kx, oc)
// (S > 1: the space-to-depth filters, as hip_conv_nhwc_xpose_filts: tap (a, b), channel c2 = c*S*S + dy*S + dx is tap (S*a + dy - OFY, S*b + dx - OFX) of channel c)
CUCL_GLOBAL_KERNEL void hip_conv_nhwc_xpose_filts_patch( GASQ float const * const filts_ref, GASQ xp_bf16x8_t * const filts, uint32_t const n, uint32_t const C,
                                                          uint32_t const KH, uint32_t const KW, uint32_t const OC, uint32_t const C2, uint32_t const KH2,
                                                          uint32_t const KW2, uint32_t const S, uint32_t const OFY, uint32_t const OFX ) {
  uint32_t const i = GLOB_ID_1D;
  if( i >= n ) { return; }
  uint32_t const oc = i % OC, b = ( i / OC ) % KW2, a = ( i / ( OC*KW2 ) ) % KH2, g = i / ( OC*KW2*KH2 );
  xp_bf16x8_t r;
  for( uint32_t e = 0; e != 8; ++e ) {
    uint32_t const c2 = 8*g + e, c = c2 / ( S*S ), dy = ( c2 / S ) % S, dx = c2 % S;
    int32_t const y = (int32_t)( S*a + dy ) - (int32_t)OFY, x = (int32_t)( S*b + dx ) - (int32_t)OFX;
    bool const ok = ( c2 < C2 ) && ( y >= 0 ) && ( y < (int32_t)KH ) && ( x >= 0 ) && ( x < (int32_t)KW );
    r[e] = (__bf16)( ok ? filts_ref[( ( oc*C + c )*KH + y )*KW + x] : 0.0f );
  }
  filts[i] = r;
}
// out img:y:x:chan (bf16 / float) -> out_ref img:chan:y:x float
CUCL_GLOBAL_KERNEL void hip_conv_nhwc_xpose_out_bf16( GASQ __bf16 const * const out, GASQ float * const out_ref, uint32_t const n, uint32_t const C, uint32_t const HW ) {
  // CUCL IX GLOB_ID_1D out_ref
  uint32_t const i = GLOB_ID_1D;
  if( i >= n ) { return; }
  uint32_t const pel = i % HW, c = ( i / HW ) % C, img = i / ( C * HW );
  out_ref[i] = (float)out[( img*HW + pel )*C + c];
}
CUCL_GLOBAL_KERNEL void hip_conv_nhwc_xpose_out_f32( GASQ float const * const out, GASQ float * const out_ref, uint32_t const n, uint32_t const C, uint32_t const HW ) {
  // CUCL IX GLOB_ID_1D out_ref
  uint32_t const i = GLOB_ID_1D;
  if( i >= n ) { return; }
  uint32_t const pel = i % HW, c = ( i / HW ) % C, img = i / ( C * HW );
  out_ref[i] = out[( img*HW + pel )*C + c];
}
"""
XPOSE_FUNCS: Dict[str, List[str]] = {
    "hip_conv_nhwc_xpose_in": ["in_ref", "in", "n", "C", "H", "W", "C2", "C8", "H2", "W2", "S", "PRY", "PRX"],
    "hip_conv_nhwc_xpose_filts": ["filts_ref", "filts", "n", "C", "KH", "KW", "C2", "C8", "KH2", "KW2", "S", "OFY", "OFX"],
    "hip_conv_nhwc_xpose_filts_patch": ["filts_ref", "filts", "n", "C", "KH", "KW", "OC", "C2", "KH2", "KW2", "S", "OFY", "OFX"],
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


def _spec_compile(rtc, name: str, src: str, subst: Dict[str, object], args: List[str]) -> None:
    done = rtc.__dict__.setdefault("_nhwc_spec_compiled", set())
    if name in done:
        return
    text = src.replace("@NAME@", name)
    for k, v in subst.items():
        text = text.replace("@" + k + "@", str(v))
    rtc.compile([RtcFuncInfo(name, text, args, Op({"type": "fwd", "func_name": name}, {}))])
    done.add(name)


# The net-input layout pass specialised per geometry (the one layout pass INSIDE a forward step; filters are laid out once at init): one thread per chunk as the
# generic kernel, sizes literal -- the index arithmetic is multiply-shift and the eight gathers are independent, unconditional loads (clamped address + select).
XPOSE_IN_SPEC_SRC = """
typedef __bf16 xps_bf16x8_t __attribute__((ext_vector_type(8)));
CUCL_GLOBAL_KERNEL void @NAME@( GASQ float const * const in_ref, GASQ xps_bf16x8_t * const in, uint32_t const n ) {
  // CUCL IX GLOB_ID_1D in n=n
  uint32_t const i = GLOB_ID_1D;
  if( i >= n ) { return; }
  uint32_t const q = i % @C8@u, p = i / @C8@u, X = p % @W2@u, r = p / @W2@u, Y = r % @H2@u, img = r / @H2@u;
  GASQ float const * const base = in_ref + (size_t)img*( @C@u*@H@u*@W@u );
  float f[8];
#pragma unroll
  for( uint32_t e = 0; e != 8; ++e ) {
    uint32_t const c2 = 8*q + e, c = c2 / ( @S@u*@S@u ), dy = ( c2 / @S@u ) % @S@u, dx = c2 % @S@u;
    int32_t const y = (int32_t)( @S@u*Y + dy ) - @PRY@, x = (int32_t)( @S@u*X + dx ) - @PRX@;
    bool const ok = ( c2 < @C2@u ) && ( y >= 0 ) && ( y < @H@ ) && ( x >= 0 ) && ( x < @W@ );
    int32_t const yc = ( y < 0 ) ? 0 : ( ( y >= @H@ ) ? @H@ - 1 : y ), xc = ( x < 0 ) ? 0 : ( ( x >= @W@ ) ? @W@ - 1 : x );
    uint32_t const cc = ( c < @C@u ) ? c : 0u;
    float const v = base[( cc*@H@u + (uint32_t)yc )*@W@u + (uint32_t)xc];
    f[e] = ok ? v : 0.0f;
  }
  xps_bf16x8_t o;
  for( uint32_t e = 0; e != 8; ++e ) { o[e] = (__bf16)f[e]; }
  in[i] = o;
}
"""


def xpose_call(arg: str, ref_vn: str, vn: str, ref_dims: Dims, dims: Dims, anno: Op = None, rtc=None) -> RtcFuncCall:
    """The layout pass between `<arg>_ref` (reference layout, float) and `<arg>` (kernel layout): in / filts forward, out backward.  `anno`:
    the annotated op (its nhwc_s2d scalars select the space-to-depth form of in / filts).  With `rtc` (arg "in"): the geometry-specialised kernel."""
    s = pry = prx = ofy = ofx = 0
    if anno is not None and anno.has("nhwc_s2d"):
        s, pry, prx = anno.get_u32("nhwc_s2d"), anno.get_u32("nhwc_s2d_pry"), anno.get_u32("nhwc_s2d_prx")
        pad = anno.get_dims("in_pad_ref"); ofy, ofx = pry - pad.dsz("y"), prx - pad.dsz("x")
    if arg == "in":
        n = dims.dims_prod() // 8
        C = ref_dims.dsz("chan")
        am = {"in_ref": RtcArg.var(ref_vn), "in": RtcArg.var(vn), "n": _u32(n), "C": _u32(C), "H": _u32(ref_dims.dsz("y")), "W": _u32(ref_dims.dsz("x")),
              "C2": _u32(C * (s * s if s else 1)), "C8": _u32(dims.dsz("chan") // 8), "H2": _u32(dims.dsz("y")), "W2": _u32(dims.dsz("x")), "S": _u32(s or 1),
              "PRY": _u32(pry), "PRX": _u32(prx)}
        if rtc is not None and ref_dims.dims_prod() < (1 << 31):
            sub = {"C": C, "H": ref_dims.dsz("y"), "W": ref_dims.dsz("x"), "C2": C * (s * s if s else 1), "C8": dims.dsz("chan") // 8, "H2": dims.dsz("y"), "W2": dims.dsz("x"),
                   "S": s or 1, "PRY": pry, "PRX": prx}
            name = "nhwc_xpose_in_" + "_".join(f"{k.lower()}{v}" for k, v in sub.items())
            _spec_compile(rtc, name, XPOSE_IN_SPEC_SRC, sub, ["in_ref", "in", "n"])
            return RtcFuncCall(name, {"in_ref": RtcArg.var(ref_vn), "in": RtcArg.var(vn), "n": _u32(n)}, tpb=_TPB, blks=(n + _TPB - 1) // _TPB)
        return RtcFuncCall("hip_conv_nhwc_xpose_in", am, tpb=_TPB, blks=(n + _TPB - 1) // _TPB)
    if arg == "filts" and dims.has("in_grp"):     # F' for the input-patch kernel
        n = dims.dims_prod() // 8
        C = ref_dims.dsz("in_chan")
        am = {"filts_ref": RtcArg.var(ref_vn), "filts": RtcArg.var(vn), "n": _u32(n), "C": _u32(C), "KH": _u32(ref_dims.dsz("y")),
              "KW": _u32(ref_dims.dsz("x")), "OC": _u32(ref_dims.dsz("out_chan")), "C2": _u32(C * (s * s if s else 1)), "KH2": _u32(dims.dsz("y")), "KW2": _u32(dims.dsz("x")),
              "S": _u32(s or 1), "OFY": _u32(ofy), "OFX": _u32(ofx)}
        return RtcFuncCall("hip_conv_nhwc_xpose_filts_patch", am, tpb=_TPB, blks=(n + _TPB - 1) // _TPB)
    if arg == "filts":
        n = dims.dims_prod() // 8
        C = ref_dims.dsz("in_chan")
        am = {"filts_ref": RtcArg.var(ref_vn), "filts": RtcArg.var(vn), "n": _u32(n), "C": _u32(C), "KH": _u32(ref_dims.dsz("y")), "KW": _u32(ref_dims.dsz("x")),
              "C2": _u32(C * (s * s if s else 1)), "C8": _u32(dims.dsz("in_chan") // 8), "KH2": _u32(dims.dsz("y")), "KW2": _u32(dims.dsz("x")), "S": _u32(s or 1),
              "OFY": _u32(ofy), "OFX": _u32(ofx)}
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
  // CUCL IX GLOB_ID_1D out n=n
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
// across-channel LRN (test/rtc/lrn.cucl): out[c] = in[c] * ( k + alpha/local_size * sum_{|d| <= local_size/2} in[c+d]^2 ) ^ -beta.  One thread per
// 16-byte chunk (8 channels) of one position; consecutive lanes hold consecutive chunks, so the halo -- the last `half` channels of the chunk
// below and the first `half` of the chunk above (half = local_size/2 <= 8) -- comes from the neighbouring LANES (wave shuffles of the squares)
// instead of from memory: every element is loaded once.  Lanes at a position's first / last chunk, and at the wave's edges, take zero / reload.
CUCL_GLOBAL_KERNEL void nhwc_lrn( GASQ bf16x8_t const * const in, GASQ bf16x8_t * const out, uint32_t const n, uint32_t const C8, uint32_t const local_size,
                                  float const alpha, float const beta, float const k ) {
  // CUCL IX GLOB_ID_1D out n=n wave_local
  uint32_t const i = GLOB_ID_1D;
  bool const live = i < n;
  int32_t const q = live ? (int32_t)( i % C8 ) : 0, half = local_size / 2, lane = LOC_ID_1D & 63;
  float v[8], sq[24];
  bf16x8_t const mid = live ? in[i] : (bf16x8_t)0;
  for( int32_t e = 0; e != 8; ++e ) { v[e] = (float)mid[e]; sq[8 + e] = v[e]*v[e]; }
  for( int32_t e = 0; e != 8; ++e ) {          // squares of the chunk below (lane - 1) and above (lane + 1)
    sq[e] = __shfl_up( sq[8 + e], 1, 64 );
    sq[16 + e] = __shfl_down( sq[8 + e], 1, 64 );
  }
  bool const has_lo = q > 0, has_hi = q + 1 < (int32_t)C8;
  if( has_lo && lane == 0 ) { bf16x8_t const lo = in[i - 1]; for( int32_t e = 0; e != 8; ++e ) { float const f = (float)lo[e]; sq[e] = f*f; } }
  if( has_hi && live && ( lane == 63 || i + 1 >= n ) ) { bf16x8_t const hi = in[i + 1]; for( int32_t e = 0; e != 8; ++e ) { float const f = (float)hi[e]; sq[16 + e] = f*f; } }
  if( !has_lo ) { for( int32_t e = 0; e != 8; ++e ) { sq[e] = 0.0f; } }
  if( !has_hi ) { for( int32_t e = 0; e != 8; ++e ) { sq[16 + e] = 0.0f; } }
  if( !live ) { return; }
  float const per_elem = alpha / (float)local_size;
  bf16x8_t r;
  for( int32_t e = 0; e != 8; ++e ) {
    float sumsq = 0.0f;
    for( int32_t d = -8; d <= 8; ++d ) { if( d >= -half && d <= half ) { sumsq += sq[8 + e + d]; } }
    r[e] = (__bf16)( v[e] * powf( k + sumsq * per_elem, -beta ) );
  }
  out[i] = r;
}
// stand-alone ReLU (one that could not be fused into its conv)
CUCL_GLOBAL_KERNEL void nhwc_relu( GASQ __bf16 * const inout, uint32_t const n ) {
  // CUCL IX GLOB_ID_1D inout n=n
  uint32_t const i = GLOB_ID_1D;
  if( i < n ) { if( (float)inout[i] <= 0.0f ) { inout[i] = (__bf16)0.0f; } }
}
// Concat: copy one input (C8_in chunks per position) into its channel range of the output (src/rtc_fwd.cc:267-280)
CUCL_GLOBAL_KERNEL void nhwc_copy( GASQ bf16x8_t const * const in, GASQ bf16x8_t * const out, uint32_t const n, uint32_t const C8_in, uint32_t const C8_out,
                                   uint32_t const off8 ) {
  // CUCL IX GLOB_ID_1D in n=n
  uint32_t const i = GLOB_ID_1D;
  if( i >= n ) { return; }
  uint32_t const pel = i / C8_in;
  out[pel*C8_out + off8 + ( i - pel*C8_in )] = in[i];
}
"""
FWD_FUNCS: Dict[str, List[str]] = {
    "nhwc_pool": ["in", "out", "n", "C8", "H", "W", "OH", "OW", "KH", "KW", "SY", "SX", "PY", "PX", "avg_pool"],
    "nhwc_lrn": ["in", "out", "n", "C8", "local_size", "alpha", "beta", "k"],
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


# Geometry-specialised forms of nhwc_pool / nhwc_lrn (the backend compiles at run time anyway -- the reference instantiates pool.cucl / lrn.cucl per
# geometry the same way, src/rtc_func_gen.cc): window, stride, padding and plane sizes are literals, so the taps unroll into independent 16-byte loads
# issued back to back (a tap outside the plane is loaded from the nearest position INSIDE the clipped window -- a duplicate does not change a maximum;
# an average adds it as zero), index arithmetic is multiply-shift, LRN shuffles only the `half` squares it needs from each neighbour lane and takes
# x^-beta as exp2(-beta * log2 x) (v_log_f32 / v_exp_f32, relative error ~3e-7 against a result stored with 8 bits of mantissa).  Same values as the
# generic kernels for max pooling (exact) and the same order of additions for averages; LRN within one bf16 rounding (tests/test_gpu_fullnet.py).
POOL_SPEC_SRC = """
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
// one thread: XT consecutive outputs along x of one (img, oy, 8 channels); the (XT-1)*SX + KW input columns they touch are loaded once ( x KH rows)
CUCL_GLOBAL_KERNEL void @NAME@( GASQ bf16x8_t const * const in, GASQ bf16x8_t * const out, uint32_t const n ) {
  // CUCL IX GLOB_ID_1D out n=n
  uint32_t const i = GLOB_ID_1D;
  if( i >= n ) { return; }
  uint32_t const c = i % @C8@u, p = i / @C8@u, xg = p % @OWG@u, q = p / @OWG@u, oy = q % @OH@u, img = q / @OH@u;
  int32_t const y0 = (int32_t)( oy*@SY@u ) - @PY@, xs = (int32_t)( xg*( @XT@u*@SX@u ) ) - @PX@;
  int32_t const ya = ( y0 < 0 ) ? 0 : y0, yb = ( y0 + @KH@ > @H@ ) ? @H@ : y0 + @KH@;
  GASQ bf16x8_t const * const base = in + (size_t)img*( @H@u*@W@u*@C8@u ) + c;
  bf16x8_t v[@NCOL@][@KH@];
#pragma unroll
  for( int32_t j = 0; j != @NCOL@; ++j ) {
#pragma unroll
    for( int32_t ky = 0; ky != @KH@; ++ky ) {
      int32_t const y = y0 + ky, x = xs + j;
      int32_t const yc = ( y < ya ) ? ya : ( ( y >= yb ) ? yb - 1 : y ), xc = ( x < 0 ) ? 0 : ( ( x >= @W@ ) ? @W@ - 1 : x );
      v[j][ky] = base[( yc*@W@ + xc )*@C8@];
    }
  }
#pragma unroll
  for( int32_t t = 0; t != @XT@; ++t ) {
    uint32_t const ox = xg*@XT@u + t;
    int32_t const x0 = xs + t*@SX@;
    int32_t const xa = ( x0 < 0 ) ? 0 : x0, xb = ( x0 + @KW@ > @W@ ) ? @W@ : x0 + @KW@;
    float acc[8];
    for( int32_t e = 0; e != 8; ++e ) { acc[e] = @AVG@ ? 0.0f : -FLT_MAX; }
#pragma unroll
    for( int32_t kx = 0; kx != @KW@; ++kx ) {
#pragma unroll
      for( int32_t ky = 0; ky != @KH@; ++ky ) {
        bool const ok = ( y0 + ky >= ya ) && ( y0 + ky < yb ) && ( x0 + kx >= xa ) && ( x0 + kx < xb );
        for( int32_t e = 0; e != 8; ++e ) {
          #pragma clang fp reassociate(off) contract(off)      // (CUCL sources build with fast-math: an average keeps the written order of its additions)
          float const f = (float)v[t*@SX@ + kx][ky][e];
          if( @AVG@ ) { if( ok ) { acc[e] = acc[e] + f; } } else { acc[e] = ( f > acc[e] ) ? f : acc[e]; }
        }
      }
    }
    float const area = (float)( ( yb - ya ) * ( xb - xa ) );
    bf16x8_t r;
    for( int32_t e = 0; e != 8; ++e ) { r[e] = (__bf16)( @AVG@ ? acc[e] / area : acc[e] ); }
    if( ( @OW@ % @XT@ == 0 ) || ( ox < @OW@u ) ) { out[( ( img*@OH@u + oy )*@OW@u + ox )*@C8@u + c] = r; }
  }
}
"""
LRN_SPEC_SRC = """
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
CUCL_GLOBAL_KERNEL void @NAME@( GASQ bf16x8_t const * const in, GASQ bf16x8_t * const out, uint32_t const n, float const alpha, float const beta, float const k ) {
  // CUCL IX GLOB_ID_1D out n=n wave_local
  uint32_t const i = GLOB_ID_1D;
  bool const live = i < n;
  int32_t const q = live ? (int32_t)( i % @C8@u ) : 0, lane = LOC_ID_1D & 63;
  float v[8], sq[8 + 2*@HALF@];     // squares of channels [-HALF, 8 + HALF) relative to this chunk
  bf16x8_t const mid = live ? in[i] : (bf16x8_t)0;
  for( int32_t e = 0; e != 8; ++e ) { v[e] = (float)mid[e]; sq[@HALF@ + e] = v[e]*v[e]; }
  for( int32_t e = 0; e != @HALF@; ++e ) {   // the last HALF squares of the chunk below (lane - 1), the first HALF of the chunk above (lane + 1)
    sq[e] = __shfl_up( sq[@HALF@ + 8 - @HALF@ + e], 1, 64 );
    sq[@HALF@ + 8 + e] = __shfl_down( sq[@HALF@ + e], 1, 64 );
  }
  bool const has_lo = q > 0, has_hi = q + 1 < @C8@;
  if( has_lo && lane == 0 ) { bf16x8_t const lo = in[i - 1]; for( int32_t e = 0; e != @HALF@; ++e ) { float const f = (float)lo[8 - @HALF@ + e]; sq[e] = f*f; } }
  if( has_hi && live && ( lane == 63 || i + 1 >= n ) ) { bf16x8_t const hi = in[i + 1]; for( int32_t e = 0; e != @HALF@; ++e ) { float const f = (float)hi[e]; sq[@HALF@ + 8 + e] = f*f; } }
  if( !has_lo ) { for( int32_t e = 0; e != @HALF@; ++e ) { sq[e] = 0.0f; } }
  if( !has_hi ) { for( int32_t e = 0; e != @HALF@; ++e ) { sq[@HALF@ + 8 + e] = 0.0f; } }
  if( !live ) { return; }
  float const per_elem = alpha / @LOCAL_SIZE@.0f;
  bf16x8_t r;
  for( int32_t e = 0; e != 8; ++e ) {
    float sumsq = 0.0f;
    for( int32_t d = 0; d != 2*@HALF@ + 1; ++d ) { sumsq += sq[e + d]; }        // ascending channel order, as the generic kernel
    r[e] = (__bf16)( v[e] * __builtin_amdgcn_exp2f( -beta * __builtin_amdgcn_logf( k + sumsq * per_elem ) ) );
  }
  out[i] = r;
}
"""


# Max pooling and across-channel LRN as ONE pass over the tensor (round 4b): GoogLeNet's pool1 -> norm1 and norm2 -> pool2, AlexNet's norm1 -> pool1 and norm2 -> pool2
# all run at the HBM roof by themselves (5.0-5.5 TB/s), so the only thing left to save is the intermediate tensor's write + read.  One thread = XT consecutive outputs
# along x of one (img, oy, 8-channel chunk), as in the pooling kernel; for every window position it loads its own chunk AND the two neighbouring chunks (the LRN window
# reaches HALF channels into each: L1 hits, the neighbouring threads load them as their own).  @LRN_FIRST@ = 1: LRN of every window position (rounded to bf16 as the LRN
# kernel stores it), then the maximum; 0: maximum of the 8 + 2 HALF channels (rounded to bf16 as the pooling kernel stores it), then LRN.  Same expressions, same order of
# additions, same roundings as the two kernels run apart: bit-identical output.  Window positions outside the plane are clamped onto it (maximum: duplicates are harmless).
POOL_LRN_SPEC_SRC = """
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
CUCL_GLOBAL_KERNEL __launch_bounds__(256) void @NAME@( GASQ bf16x8_t const * const in, GASQ bf16x8_t * const out, uint32_t const n, float const alpha, float const beta, float const k ) {
  // CUCL IX GLOB_ID_1D out n=n
  uint32_t const i = GLOB_ID_1D;
  if( i >= n ) { return; }
  uint32_t const c = i % @C8@u, p = i / @C8@u, xg = p % @OWG@u, q = p / @OWG@u, oy = q % @OH@u, img = q / @OH@u;
  int32_t const y0 = (int32_t)( oy*@SY@u ) - @PY@, xs = (int32_t)( xg*( @XT@u*@SX@u ) ) - @PX@;
  int32_t const ya = ( y0 < 0 ) ? 0 : y0, yb = ( y0 + @KH@ > @H@ ) ? @H@ : y0 + @KH@;
  GASQ bf16x8_t const * const base = in + (size_t)img*( @H@u*@W@u*@C8@u ) + c;
  bool const has_lo = c > 0, has_hi = c + 1 < @C8@u;
  int32_t const dlo = has_lo ? -1 : 0, dhi = has_hi ? 1 : 0;      // (a missing neighbour re-reads the chunk itself: in range, and its values are zeroed below)
  float const per_elem = alpha / @LOCAL_SIZE@.0f;
  float acc[@XT@][8 + 2*@HALF@];       // running maxima: channels [-HALF, 8 + HALF) of this chunk (LRN first: only [HALF, HALF + 8) are used)
  for( int32_t t = 0; t != @XT@; ++t ) { for( int32_t e = 0; e != 8 + 2*@HALF@; ++e ) { acc[t][e] = -FLT_MAX; } }
  // one window COLUMN at a time: its KH x 3 chunks are loaded while the previous column is worked on (all of a thread's window at once does not fit the registers)
  bf16x8_t cur[@KH@][3], nxt[@KH@][3];
#define PL_LOAD( DST, J ) { int32_t const x = xs + (J), xc = ( x < 0 ) ? 0 : ( ( x >= @W@ ) ? @W@ - 1 : x ); \
  _Pragma("unroll") for( int32_t ky = 0; ky != @KH@; ++ky ) { int32_t const y = y0 + ky, yc = ( y < ya ) ? ya : ( ( y >= yb ) ? yb - 1 : y ); \
    GASQ bf16x8_t const * const at = base + ( yc*@W@ + xc )*@C8@; DST[ky][0] = at[dlo]; DST[ky][1] = at[0]; DST[ky][2] = at[dhi]; } }
  PL_LOAD( cur, 0 );
#pragma unroll
  for( int32_t j = 0; j != @NCOL@; ++j ) {
    if( j + 1 != @NCOL@ ) { PL_LOAD( nxt, j + 1 ); }
    __builtin_amdgcn_sched_barrier( 0 );
#pragma unroll
    for( int32_t ky = 0; ky != @KH@; ++ky ) {
      float w[8 + 2*@HALF@];           // the position's channels [-HALF, 8 + HALF): zeros past the tensor's channels (their squares add nothing; as maxima they are never used)
      for( int32_t e = 0; e != @HALF@; ++e ) { w[e] = has_lo ? (float)cur[ky][0][8 - @HALF@ + e] : 0.0f; w[@HALF@ + 8 + e] = has_hi ? (float)cur[ky][2][e] : 0.0f; }
      for( int32_t e = 0; e != 8; ++e ) { w[@HALF@ + e] = (float)cur[ky][1][e]; }
#if @LRN_FIRST@
      float sq[8 + 2*@HALF@], r8[8];
      for( int32_t e = 0; e != 8 + 2*@HALF@; ++e ) { sq[e] = w[e]*w[e]; }
      for( int32_t e = 0; e != 8; ++e ) {
        float sumsq = 0.0f;
        for( int32_t d = 0; d != 2*@HALF@ + 1; ++d ) { sumsq += sq[e + d]; }        // ascending channel order, as the LRN kernels
        r8[e] = (float)(__bf16)( w[@HALF@ + e] * __builtin_amdgcn_exp2f( -beta * __builtin_amdgcn_logf( k + sumsq * per_elem ) ) );
      }
      for( int32_t e = 0; e != 8; ++e ) { w[@HALF@ + e] = r8[e]; }
#endif
#pragma unroll
      for( int32_t t = 0; t != @XT@; ++t ) {
        if( ( j >= t*@SX@ ) && ( j < t*@SX@ + @KW@ ) ) {
          for( int32_t e = ( @LRN_FIRST@ ? @HALF@ : 0 ); e != ( @LRN_FIRST@ ? @HALF@ + 8 : 8 + 2*@HALF@ ); ++e ) { acc[t][e] = ( w[e] > acc[t][e] ) ? w[e] : acc[t][e]; }
        }
      }
    }
    __builtin_amdgcn_sched_barrier( 0 );
#pragma unroll
    for( int32_t ky = 0; ky != @KH@; ++ky ) { cur[ky][0] = nxt[ky][0]; cur[ky][1] = nxt[ky][1]; cur[ky][2] = nxt[ky][2]; }
  }
#pragma unroll
  for( int32_t t = 0; t != @XT@; ++t ) {
    uint32_t const ox = xg*@XT@u + t;
    bf16x8_t r;
#if @LRN_FIRST@
    for( int32_t e = 0; e != 8; ++e ) { r[e] = (__bf16)acc[t][@HALF@ + e]; }
#else
    float pv[8 + 2*@HALF@], sq[8 + 2*@HALF@];
    for( int32_t e = 0; e != 8 + 2*@HALF@; ++e ) {
      bool const real = ( e >= @HALF@ || has_lo ) && ( e < @HALF@ + 8 || has_hi );
      pv[e] = real ? (float)(__bf16)acc[t][e] : 0.0f; sq[e] = pv[e]*pv[e];       // the pooled value as the pooling kernel stores it
    }
    for( int32_t e = 0; e != 8; ++e ) {
      float sumsq = 0.0f;
      for( int32_t d = 0; d != 2*@HALF@ + 1; ++d ) { sumsq += sq[e + d]; }
      r[e] = (__bf16)( pv[@HALF@ + e] * __builtin_amdgcn_exp2f( -beta * __builtin_amdgcn_logf( k + sumsq * per_elem ) ) );
    }
#endif
    if( ( @OW@ % @XT@ == 0 ) || ( ox < @OW@u ) ) { out[( ( img*@OH@u + oy )*@OW@u + ox )*@C8@u + c] = r; }
  }
#undef PL_LOAD
}
"""


# LRN -> max pooling through LDS (round 4c): the thread-per-output kernel above evaluates the LRN once per WINDOW POSITION (3x3 / stride 2, four outputs per thread: 6.75
# evaluations per output where the LRN kernel does 4) and is compute-bound behind v_log_f32 / v_exp_f32.  Here a workgroup owns TY output rows of one image: it normalises
# the (TY - 1) SY + KH input rows they need ONCE into LDS -- the LRN kernel's own code, chunk by chunk, halo squares from the neighbouring lanes -- and pools from LDS.
# 1 + (KH - SY) / (TY SY) of the LRN kernel's arithmetic (1.5 for the TY = 1 the planner takes: see lrn_pool_lds_rows), one read of the input, one write of the pooled output.  Same expressions, same order, same
# bf16 rounding between the two ops as the two kernels run apart: bit-identical.  A workgroup kernel: single-device backends only (the multi-device backend shards
# per-element functions), ConvPipeFwd keeps the pair apart elsewhere.
LRN_POOL_LDS_SRC = """
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
CUCL_GLOBAL_KERNEL __launch_bounds__(@TPB@) void @NAME@( GASQ bf16x8_t const * const in, GASQ bf16x8_t * const out, uint32_t const n, float const alpha, float const beta, float const k ) {
  // CUCL IX GRP_ID_1D in n=n
  // (a workgroup = one (img, output-row group): groups enumerate images batch-major, n / TPB of them -- a multi-device backend shards them by img, csrc/hip_multi.cc)
  LOCSHAR_MEM bf16x8_t lds[@NROWS@*@W@*@C8@];
  uint32_t const blk = GRP_ID_1D, tid = LOC_ID_1D;
  if( blk*@TPB@u >= n ) { return; }
  uint32_t const img = blk / @RG@u, rg = blk % @RG@u, oy0 = rg*@TY@u;
  int32_t const oy_end = ( oy0 + @TY@u < @OH@u ) ? (int32_t)oy0 + @TY@ : @OH@;      // (exclusive)
  int32_t const y_lo0 = (int32_t)( oy0*@SY@u ) - @PY@, y_lo = ( y_lo0 < 0 ) ? 0 : y_lo0;
  int32_t const y_hi0 = ( oy_end - 1 )*@SY@ - @PY@ + @KH@, y_hi = ( y_hi0 > @H@ ) ? @H@ : y_hi0;
  uint32_t const nch = (uint32_t)( y_hi - y_lo )*( @W@u*@C8@u );                    // chunks to normalise: whole input rows y_lo .. y_hi - 1, contiguous in memory
  GASQ bf16x8_t const * const src = in + ( (size_t)img*@H@u + (uint32_t)y_lo )*( @W@u*@C8@u );
  float const per_elem = alpha / @LOCAL_SIZE@.0f;
  int32_t const lane = tid & 63;
  for( uint32_t i0 = 0; i0 < nch; i0 += @TPB@u ) {      // (uniform trip count: every lane of a wave takes part in the shuffles; i0 and TPB are multiples of 64: lane <-> chunk as in the LRN kernel)
    uint32_t const i = i0 + tid;
    bool const live = i < nch;
    int32_t const q = live ? (int32_t)( i % @C8@u ) : 0;
    float v[8], sq[8 + 2*@HALF@];
    bf16x8_t const mid = live ? src[i] : (bf16x8_t)0;
    for( int32_t e = 0; e != 8; ++e ) { v[e] = (float)mid[e]; sq[@HALF@ + e] = v[e]*v[e]; }
    for( int32_t e = 0; e != @HALF@; ++e ) {
      sq[e] = __shfl_up( sq[@HALF@ + 8 - @HALF@ + e], 1, 64 );
      sq[@HALF@ + 8 + e] = __shfl_down( sq[@HALF@ + e], 1, 64 );
    }
    bool const has_lo = q > 0, has_hi = q + 1 < @C8@;
    if( has_lo && live && lane == 0 ) { bf16x8_t const lo = src[i - 1]; for( int32_t e = 0; e != @HALF@; ++e ) { float const f = (float)lo[8 - @HALF@ + e]; sq[e] = f*f; } }
    if( has_hi && live && ( lane == 63 || i + 1 >= nch ) ) { bf16x8_t const hi = src[i + 1]; for( int32_t e = 0; e != @HALF@; ++e ) { float const f = (float)hi[e]; sq[@HALF@ + 8 + e] = f*f; } }
    if( !has_lo ) { for( int32_t e = 0; e != @HALF@; ++e ) { sq[e] = 0.0f; } }
    if( !has_hi ) { for( int32_t e = 0; e != @HALF@; ++e ) { sq[@HALF@ + 8 + e] = 0.0f; } }
    if( live ) {
      bf16x8_t r;
      for( int32_t e = 0; e != 8; ++e ) {
        float sumsq = 0.0f;
        for( int32_t d = 0; d != 2*@HALF@ + 1; ++d ) { sumsq += sq[e + d]; }        // ascending channel order, as the LRN kernels
        r[e] = (__bf16)( v[e] * __builtin_amdgcn_exp2f( -beta * __builtin_amdgcn_logf( k + sumsq * per_elem ) ) );
      }
      lds[i] = r;
    }
  }
  BARRIER_SYNC;
  uint32_t const nout = (uint32_t)( oy_end - (int32_t)oy0 )*( @OW@u*@C8@u );
  for( uint32_t o = tid; o < nout; o += @TPB@u ) {
    uint32_t const c = o % @C8@u, p = o / @C8@u, ox = p % @OW@u, oy = oy0 + p / @OW@u;
    int32_t const y0 = (int32_t)( oy*@SY@u ) - @PY@, x0 = (int32_t)( ox*@SX@u ) - @PX@;
    int32_t const ya = ( y0 < 0 ) ? 0 : y0, yb = ( y0 + @KH@ > @H@ ) ? @H@ : y0 + @KH@, xa = ( x0 < 0 ) ? 0 : x0, xb = ( x0 + @KW@ > @W@ ) ? @W@ : x0 + @KW@;
    float acc[8];
    for( int32_t e = 0; e != 8; ++e ) { acc[e] = -FLT_MAX; }
#pragma unroll
    for( int32_t ky = 0; ky != @KH@; ++ky ) {
#pragma unroll
      for( int32_t kx = 0; kx != @KW@; ++kx ) {      // window positions outside the plane are clamped onto it (a maximum: duplicates are harmless)
        int32_t const y = y0 + ky, x = x0 + kx;
        int32_t const yc = ( y < ya ) ? ya : ( ( y >= yb ) ? yb - 1 : y ), xc = ( x < xa ) ? xa : ( ( x >= xb ) ? xb - 1 : x );
        bf16x8_t const w = lds[( ( yc - y_lo )*@W@ + xc )*@C8@ + c];
        for( int32_t e = 0; e != 8; ++e ) { float const f = (float)w[e]; acc[e] = ( f > acc[e] ) ? f : acc[e]; }
      }
    }
    bf16x8_t r;
    for( int32_t e = 0; e != 8; ++e ) { r[e] = (__bf16)acc[e]; }
    out[( ( img*@OH@u + oy )*@OW@u + ox )*@C8@u + c] = r;
  }
}
"""
_LRN_POOL_LDS_MAX = 112 * 1024     # bytes of LDS a workgroup may take (one workgroup of 1024 threads per CU)


def lrn_pool_lds_rows(i: Dims, o: Dims, kern, stride) -> int:
    """Output rows per workgroup (TY) of the LDS form of LRN -> Pooling: the largest of 4 / 2 / 1 whose normalised input rows fit the LDS budget; 0: none fits."""
    W, c8, OH = i.dsz("x"), i.dsz("chan") // 8, o.dsz("y")
    lim = int(os.environ.get("BODAHIP_LRN_POOL_LDS_MAX", _LRN_POOL_LDS_MAX))      # (experiments: a smaller budget gives fewer rows per workgroup, more workgroups per CU)
    # measured (MI355X, bench.py *-net at 64 / 256 images, us for the pair; apart: LRN + pooling kernels): GoogLeNet norm2 -> pool2 (56 x 56 x 192) apart 52 | TY 1 / 2: 41 / 50;
    # AlexNet norm1 -> pool1 (55 x 55 x 96) apart 96 | TY 1 / 2 / 4: 69 / 73 / 84; norm2 -> pool2 (27 x 27 x 256) apart 57 | 39 / 47 / 51 -- one output row per workgroup: more, smaller
    # workgroups per CU overlap each other's load, LRN and pooling phases, which is worth more than the rows a taller tile would not normalise twice
    forced = int(os.environ.get("BODAHIP_LRN_POOL_TY", "0"))
    for ty in ((forced,) if forced else (1,)):
        if ty <= max(1, OH) and ((ty - 1) * stride[0] + kern[0]) * W * c8 * 16 <= lim:
            return ty
    return 0


def lrn_pool_lds_call(in_vn: str, out_vn: str, i: Dims, o: Dims, kern, stride, pad, local_size: int, alpha: float, beta: float, k: float, rtc) -> RtcFuncCall:
    """LRN -> Pooling(max) through LDS (LRN_POOL_LDS_SRC).  i: the LRN's input dims, o: the pooling's output dims (channels-last)."""
    c8 = i.dsz("chan") // 8
    H, W, OH, OW = i.dsz("y"), i.dsz("x"), o.dsz("y"), o.dsz("x")
    ty = lrn_pool_lds_rows(i, o, kern, stride)
    if not ty:
        raise UnsupErr("channels-last LRN -> Pooling through LDS: the input rows of one output row do not fit the LDS")
    nrows, rg = (ty - 1) * stride[0] + kern[0], (OH + ty - 1) // ty
    tpb = 1024 if nrows * W * c8 >= 3072 else 512      # (threads per workgroup by the chunks it normalises: GoogLeNet norm2 4032 chunks 41 us on 1024 threads, 50 on 512; AlexNet norm1 1980 chunks 76 / 69)
    if os.environ.get("BODAHIP_LRN_POOL_TPB"):
        tpb = int(os.environ["BODAHIP_LRN_POOL_TPB"])
    blks = o.dsz("img") * rg
    name = f"nhwc_lrn_pool_lds_c{c8}_{H}x{W}_{OH}x{OW}_k{kern[0]}x{kern[1]}_s{stride[0]}x{stride[1]}_p{pad[0]}x{pad[1]}_n{local_size}_y{ty}_t{tpb}"
    _spec_compile(rtc, name, LRN_POOL_LDS_SRC, {"C8": c8, "H": H, "W": W, "OH": OH, "OW": OW, "KH": kern[0], "KW": kern[1], "SY": stride[0], "SX": stride[1], "PY": pad[0], "PX": pad[1],
                                                 "TY": ty, "NROWS": nrows, "RG": rg, "TPB": tpb, "HALF": local_size // 2, "LOCAL_SIZE": local_size}, ["in", "out", "n", "alpha", "beta", "k"])
    return RtcFuncCall(name, {"in": RtcArg.var(in_vn), "out": RtcArg.var(out_vn), "n": _u32(blks * tpb), "alpha": _f32(alpha), "beta": _f32(beta), "k": _f32(k)}, tpb=tpb, blks=blks)


def pool_lrn_fusable(i: Dims, o: Dims, kern, stride, pad, avg: int, local_size: int, alpha: float, k: float) -> bool:
    """Can a max pooling (input dims i, output dims o, channels-last) and an across-channel LRN run as one kernel?  The conditions of the two specialised kernels."""
    H, W, OH, OW = i.dsz("y"), i.dsz("x"), o.dsz("y"), o.dsz("x")
    nonempty = all(min(H, oy * stride[0] - pad[0] + kern[0]) > max(0, oy * stride[0] - pad[0]) for oy in (0, OH - 1)) and \
        all(min(W, ox * stride[1] - pad[1] + kern[1]) > max(0, ox * stride[1] - pad[1]) for ox in (0, OW - 1))
    return (not avg) and nonempty and kern[0] * kern[1] <= 25 and i.dims_prod() * 2 < (1 << 31) and i.dsz("chan") % 8 == 0 and \
        local_size % 2 == 1 and 1 <= local_size // 2 <= 4 and k > 0.0 and alpha >= 0.0


def pool_lrn_call(in_vn: str, out_vn: str, i: Dims, o: Dims, kern, stride, pad, local_size: int, alpha: float, beta: float, k: float, lrn_first: bool, rtc) -> RtcFuncCall:
    """One call for Pooling(max) -> LRN (lrn_first False) or LRN -> Pooling(max) (True).  i: the dims of the first op's input, o: of the second op's output."""
    c8 = i.dsz("chan") // 8
    H, W, OH, OW = i.dsz("y"), i.dsz("x"), o.dsz("y"), o.dsz("x")
    xt = 4 if OW >= 8 else (2 if OW >= 4 else 1)
    while xt > 1 and ((xt - 1) * stride[1] + kern[1]) * kern[0] > 36:
        xt //= 2
    ncol, owg = (xt - 1) * stride[1] + kern[1], (OW + xt - 1) // xt
    nt = o.dsz("img") * OH * owg * c8
    name = f"nhwc_{'lrn_pool' if lrn_first else 'pool_lrn'}_c{c8}_{H}x{W}_{OH}x{OW}_k{kern[0]}x{kern[1]}_s{stride[0]}x{stride[1]}_p{pad[0]}x{pad[1]}_n{local_size}_t{xt}"
    _spec_compile(rtc, name, POOL_LRN_SPEC_SRC, {"C8": c8, "H": H, "W": W, "OH": OH, "OW": OW, "KH": kern[0], "KW": kern[1], "SY": stride[0], "SX": stride[1], "PY": pad[0], "PX": pad[1],
                                                  "XT": xt, "NCOL": ncol, "OWG": owg, "HALF": local_size // 2, "LOCAL_SIZE": local_size, "LRN_FIRST": int(bool(lrn_first))},
                  ["in", "out", "n", "alpha", "beta", "k"])
    return RtcFuncCall(name, {"in": RtcArg.var(in_vn), "out": RtcArg.var(out_vn), "n": _u32(nt), "alpha": _f32(alpha), "beta": _f32(beta), "k": _f32(k)},
                       tpb=_TPB, blks=(nt + _TPB - 1) // _TPB)


def pool_call(in_vn: str, out_vn: str, i: Dims, o: Dims, kern, stride, pad, avg: int, rtc=None) -> RtcFuncCall:
    """i / o: the channels-last dims of the vars (img:y:x:chan, chan a multiple of 8).  With `rtc`: the geometry-specialised kernel (compiled on first use) where
    every window meets the plane and is small enough to unroll; the generic kernel otherwise."""
    c8 = i.dsz("chan") // 8; n = o.dsz("img") * o.dsz("y") * o.dsz("x") * c8
    H, W, OH, OW = i.dsz("y"), i.dsz("x"), o.dsz("y"), o.dsz("x")
    nonempty = all(min(H, oy * stride[0] - pad[0] + kern[0]) > max(0, oy * stride[0] - pad[0]) for oy in (0, OH - 1)) and \
        all(min(W, ox * stride[1] - pad[1] + kern[1]) > max(0, ox * stride[1] - pad[1]) for ox in (0, OW - 1))
    if rtc is not None and nonempty and kern[0] * kern[1] <= 64 and i.dims_prod() * 2 < (1 << 31):
        xt = 4 if OW >= 8 else (2 if OW >= 4 else 1)                 # outputs along x per thread: their windows share input columns (measured: 4 ahead of 1 / 2 / 8)
        while xt > 1 and ((xt - 1) * stride[1] + kern[1]) * kern[0] > 36:
            xt //= 2
        ncol, owg = (xt - 1) * stride[1] + kern[1], (OW + xt - 1) // xt
        nt = o.dsz("img") * OH * owg * c8
        name = f"nhwc_pool_c{c8}_{H}x{W}_{OH}x{OW}_k{kern[0]}x{kern[1]}_s{stride[0]}x{stride[1]}_p{pad[0]}x{pad[1]}_{'avg' if avg else 'max'}_t{xt}"
        _spec_compile(rtc, name, POOL_SPEC_SRC, {"C8": c8, "H": H, "W": W, "OH": OH, "OW": OW, "KH": kern[0], "KW": kern[1], "SY": stride[0], "SX": stride[1],
                                                 "PY": pad[0], "PX": pad[1], "AVG": int(bool(avg)), "XT": xt, "NCOL": ncol, "OWG": owg}, ["in", "out", "n"])
        return RtcFuncCall(name, {"in": RtcArg.var(in_vn), "out": RtcArg.var(out_vn), "n": _u32(nt)}, tpb=_TPB, blks=(nt + _TPB - 1) // _TPB)
    am = {"in": RtcArg.var(in_vn), "out": RtcArg.var(out_vn), "n": _u32(n), "C8": _u32(c8), "H": _u32(i.dsz("y")), "W": _u32(i.dsz("x")), "OH": _u32(o.dsz("y")),
          "OW": _u32(o.dsz("x")), "KH": _u32(kern[0]), "KW": _u32(kern[1]), "SY": _u32(stride[0]), "SX": _u32(stride[1]), "PY": _u32(pad[0]), "PX": _u32(pad[1]),
          "avg_pool": _u32(avg)}
    return RtcFuncCall("nhwc_pool", am, tpb=_TPB, blks=(n + _TPB - 1) // _TPB)


def lrn_call(in_vn: str, out_vn: str, d: Dims, local_size: int, alpha: float, beta: float, k: float, rtc=None) -> RtcFuncCall:
    if local_size // 2 > 8:
        raise UnsupErr("channels-last LRN: local_size above 17")
    n = d.dims_prod() // 8
    if rtc is not None and local_size % 2 == 1 and k > 0.0 and alpha >= 0.0:     # (k + alpha * sum of squares > 0: the log is defined)
        c8 = d.dsz("chan") // 8; name = f"nhwc_lrn_c{c8}_n{local_size}"
        _spec_compile(rtc, name, LRN_SPEC_SRC, {"C8": c8, "HALF": local_size // 2, "LOCAL_SIZE": local_size}, ["in", "out", "n", "alpha", "beta", "k"])
        return RtcFuncCall(name, {"in": RtcArg.var(in_vn), "out": RtcArg.var(out_vn), "n": _u32(n), "alpha": _f32(alpha), "beta": _f32(beta), "k": _f32(k)},
                           tpb=_TPB, blks=(n + _TPB - 1) // _TPB)
    am = {"in": RtcArg.var(in_vn), "out": RtcArg.var(out_vn), "n": _u32(n), "C8": _u32(d.dsz("chan") // 8), "local_size": _u32(local_size), "alpha": _f32(alpha),
          "beta": _f32(beta), "k": _f32(k)}
    return RtcFuncCall("nhwc_lrn", am, tpb=_TPB, blks=(n + _TPB - 1) // _TPB)


def relu_call(vn: str, d: Dims) -> RtcFuncCall:
    n = d.dims_prod()
    return RtcFuncCall("nhwc_relu", {"inout": RtcArg.var(vn), "n": _u32(n)}, tpb=_TPB, blks=(n + _TPB - 1) // _TPB)


def copy_call(in_vn: str, out_vn: str, i: Dims, o: Dims, chan_off: int) -> RtcFuncCall:
    n = i.dims_prod() // 8
    am = {"in": RtcArg.var(in_vn), "out": RtcArg.var(out_vn), "n": _u32(n), "C8_in": _u32(i.dsz("chan") // 8), "C8_out": _u32(o.dsz("chan") // 8), "off8": _u32(chan_off // 8)}
    return RtcFuncCall("nhwc_copy", am, tpb=_TPB, blks=(n + _TPB - 1) // _TPB)
