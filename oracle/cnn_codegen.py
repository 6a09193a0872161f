"""CUCL-template compatibility mode: the reference's OWN convolution / sgemm variants -- `sgemm`, `conv`, `k1conv`, `tconv`
and their `xpose_*` passes, the templates under test/rtc/ of a Boda checkout -- instantiated for `be=hip`.

Two layers of the reference are restated here (behaviour, not code):
  * blocking and layout choice: `gbt_tile_t` (src/gbt_tile.H:12-67) and the conv / sgemm branches of `add_cnn_codegen_annotations` /
    `add_codegen_annotations` (src/cnn_op.cc:16-68,142-330,338-378): which variant runs an op, the three-level `work` blocking
    (blocks x threads x per-thread tile), and the transposed dims of `in` / `filts` the variant wants (originals kept as `<arg>_ref`);
  * the variants' custom code generation (`cnn_custom_codegen_t`, src/cnn_codegen.cc:137-163,165-215,460-515,625-823): the unrolled
    load / fma / store sections the templates leave open (`%(inner_loop_body)`, `%(stores)`, ...), emitted into the template through
    `CallGen.line / set` (boda_amd/cucl_template.py).
The templates themselves are NOT part of this repository: `instantiate_ref` reads them from a Boda checkout's test/rtc directory.  Where a
checkout exists (the build container), oracle/ref_cucl.py instantiates them for a list of ops and compiles them for gfx950 into
oracle/_ref/ (code objects + a manifest of launch geometries) -- the reference's real kernels, timed and checked on the GPU box beside the
native ones (tests/test_gpu_ref_cucl.py; their times: DESIGN.md section 5).
This module is TEST INFRASTRUCTURE (it lives under oracle/ with its only users, oracle/ref_cucl.py and the tests): the product's own annotation
layer is boda_amd/cnn_op.py, which routes ops to the native kernels.

Covered: all four sgemm variants (use_local_mem 0..3; vector width 2 / 4), conv / conv_simd / k1conv (incl. write-xposed
chaining into the next k1conv) / k1conv_simd / tconv / ipconv, reduce; sgemm_prof (not instantiable in the reference either) and the backward ops raise UnsupErr.
"""
from __future__ import annotations
import os
from typing import Callable, Dict, List, Optional, Tuple

from boda_amd.cnn_op import OpTune
from boda_amd.cucl_template import CallGen, Instance, instantiate, load_template
from boda_amd.op import Dims, Nda, Op, RtErr, UnsupErr


def _cdiv(a: int, b: int) -> int:
    return -(-a // b)


# ---------------------------------------------------------------------------------------------------------------------------------------
# blocking: src/gbt_tile.H
# ---------------------------------------------------------------------------------------------------------------------------------------
def good_div(v: int, target: int) -> int:
    """`target` if splitting v into chunks of it wastes < 20 % of v, else the next smaller chunk size that does (src/gbt_tile.H:12-21)."""
    if v <= target:
        return v
    d = target
    while (_cdiv(v, d) * d - v) * 5 >= v:
        d -= 1
    return d


class GbtTile:
    """An M x N space tiled into blocks of threads of per-thread tiles (src/gbt_tile.H:25-67): mn_per_thr from the target tile (M adjusted
    to divide well, N at most 3x M), thr_per_blk = (as many M rows as max_tpb allows, min(16, 3 * M threads, N threads)), num_blk = ceil."""

    def __init__(self, t_tile: Tuple[int, int], max_tpb: int, num_mn: Tuple[int, int]):
        m_per = good_div(num_mn[0], t_tile[0])
        n_per = min(t_tile[1], m_per * 3)
        self.mn_per_thr = (m_per, n_per)
        self.num_thr = (_cdiv(num_mn[0], m_per), _cdiv(num_mn[1], n_per))
        tn = min(min(16, self.num_thr[0] * 3), self.num_thr[1])
        tm = 0
        while (tm + 1) * tn <= max_tpb:
            tm += 1
            if tm * m_per >= num_mn[0]:
                break
        if not tm:
            raise RtErr("gbt_tile: no thread fits max_tpb")
        self.thr_per_blk = (tm, tn)
        self.num_blk = (_cdiv(self.num_thr[0], tm), _cdiv(self.num_thr[1], tn))


# ---------------------------------------------------------------------------------------------------------------------------------------
# annotations: src/cnn_op.cc
# ---------------------------------------------------------------------------------------------------------------------------------------
def _none_dims(**kw: int) -> Dims:
    return Dims(tuple(kw.keys()), tuple(int(v) for v in kw.values()), "none")


def ref_conv_func_name(op: Op, tune: OpTune) -> str:
    """Variant choice of src/cnn_op.cc:46-68 (without the culibs door)."""
    g = op.conv_geom()
    if tune.ipconv and g["PY"] == 0 and g["PX"] == 0 and (g["OH"], g["OW"]) == (1, 1):
        return "ipconv"
    if tune.k1conv and (g["KH"], g["KW"]) == (1, 1) and (g["SY"], g["SX"]) == (1, 1) and 6 <= g["OW"] <= 300 and g["OC"] >= 64:
        return "conv" if (g["PY"] or g["PX"]) else ("k1conv_simd" if tune.use_local_mem == 2 else "k1conv")
    if tune.tconv and (tune.tconv == 2 or (g["KW"] <= tune.tconv_max_ksz[0] and g["KH"] <= tune.tconv_max_ksz[1] and g["OW"] >= 6)):
        return "tconv"
    return "conv_simd" if tune.use_local_mem == 2 else "conv"


def annotate_ref(op: Op, tune: OpTune) -> Op:
    """The annotated op the reference's code generator works from: func_name = the CUCL variant, `work` = the blocking, `in` / `filts` in the
    variant's layout with the originals as in_ref / filts_ref / out_ref."""
    a = op.copy()
    t = a.get_type()
    if t == "sgemm":
        g = a.sgemm_geom()
        variants = {0: "sgemm_no_local", 1: "sgemm", 2: "sgemm_simd", 3: "sgemm_simd_local"}      # src/cnn_op.cc:368-374
        if tune.use_local_mem not in variants:
            raise RtErr(f"op_tune.use_local_mem must be 0..3, got {tune.use_local_mem}")
        mb, nb = tune.MNb[0] * tune.MNt[0], tune.MNb[1] * tune.MNt[1]
        for what, v, blk in (("M", g["M"], mb), ("N", g["N"], nb), ("K", g["K"], tune.Kb)):
            if v % blk:   # (the reference's own restriction, src/cnn_op.cc:349-360: its default tune cannot run sgemm 64^3)
                raise RtErr(f"FIXME: currently, {what}={v} must be a multiple of {what}_blk={blk}")
        a.set_dims("work", _none_dims(Mg=g["M"] // mb, Ng=g["N"] // nb, Mb=tune.MNb[0], Nb=tune.MNb[1], Kb=tune.Kb, Mt=tune.MNt[0], Nt=tune.MNt[1]))
        a.set_u32("use_local_mem", tune.use_local_mem); a.set_u32("prof_variant", tune.prof_variant); a.set_u32("vw", tune.vw)
        if tune.prof_variant:
            # (src/cnn_op.cc:365-367 would name the function sgemm_prof -- a memory-access probe, c = a + b -- but test/rtc/sgemm_prof.cucl reads %(prof_variant), which
            # the reference's template layer only defines for DECLARED by-value arguments (src/rtc_func_gen.cc:160-176): the template declares none, so the reference
            # itself stops with "unknown template variable" there.  Nothing to be compatible with.)
            raise UnsupErr("CUCL compatibility mode: sgemm_prof cannot be instantiated (its template reads %(prof_variant) without declaring it, in the reference too)")
        a.set_func_name(variants[tune.use_local_mem])
        return a
    if t != "Convolution":
        raise UnsupErr(f"CUCL compatibility mode: op type {t!r}")
    g = a.conv_geom()
    fn = ref_conv_func_name(a, tune)
    if fn not in ("conv", "k1conv", "tconv", "ipconv", "k1conv_simd", "conv_simd"):
        raise UnsupErr(f"CUCL compatibility mode: variant '{fn}' is not generated")
    a.set_func_name(fn)
    a.set_u32("conv_has_relu", 1)
    ni, no, filts = a.get_dims("in"), a.get_dims("out"), a.get_dims("filts")
    a.set_dims("in_ref", ni); a.set_dims("filts_ref", filts); a.set_dims("out_ref", no)
    pels = g["B"] * g["OH"] * g["OW"]
    gbt = GbtTile(tune.MNt, tune.MNb[0] * tune.MNb[1], (pels, g["OC"]))
    (m_per, n_per), (tm, tn), (bm, bn) = gbt.mn_per_thr, gbt.thr_per_blk, gbt.num_blk
    in_dims = ni
    if fn in ("tconv", "k1conv"):
        a.set_dims("flags", Dims((), (), "uint32_t"))       # (exactly these two variants carry the debugging input)
    if fn == "tconv":
        lines = g["B"] * g["OH"]
        blk_bline, blk_bx = _cdiv(lines, tm), _cdiv(g["OW"], m_per)
        max_imgs, b_line = 0, 0
        for _ in range(blk_bline):      # images a block of tm output lines can touch
            e_line = b_line + tm - 1
            max_imgs = max(max_imgs, min(g["B"] - 1, e_line // g["OH"]) - b_line // g["OH"] + 1)
            b_line = e_line + 1
        in_lines = (tm - max_imgs) * g["SY"] + g["KH"] * max_imgs
        x_sz = (m_per - 1) * g["SX"] + g["KW"]
        in_dims = Dims(("blk_bline", "blk_bx", "blk_in_chan", "blk_y", "blk_x"), (blk_bline, blk_bx, g["C"], in_lines, x_sz), ni.tn)
        work = _none_dims(blk_bline=blk_bline, blk_bx=blk_bx, out_chan_blk=bn, blk_y=tm, out_chan_tile=tn, pels=m_per, out_chan=n_per)
    else:
        work = _none_dims(pels_blk=bm, out_chan_blk=bn, pels_tile=tm, out_chan_tile=tn, pels=m_per, out_chan=n_per)
        if fn == "k1conv":
            in_dims = Dims(("blk", "blk_iter", "blk_iter_chan", "blk_pel"), (bm, _cdiv(g["C"], tune.Kb), tune.Kb, tm * m_per), ni.tn)
    if fn == "conv_simd":      # the general vector variant (src/cnn_op.cc:246-292): input planes padded to (in + pad) rounded up to the stride, the output
        # computed on the (padded plane / stride) grid -- a superset of the real output positions -- so that in_pel = out_pel * stride is linear
        vw = tune.vw
        if vw not in (2, 4):
            raise UnsupErr(f"CUCL compatibility mode: vector width vw={vw}: HIP has float2 / float4 (the reference's default 8 is an OpenCL type)")
        if tune.Kb != 1:
            raise RtErr("conv_simd: Kb must be 1 (no inner-loop unrolling in this variant, src/cnn_op.cc:246)")
        if g["SY"] != g["SX"]:
            raise RtErr("conv_simd: uniform x / y stride only")
        a.set_u32("vw", vw); a.set_u32("Kb", tune.Kb)
        st = g["SY"]
        iy, ix = _cdiv(g["H"] + g["PY"], st) * st, _cdiv(g["W"] + g["PX"], st) * st
        in_pels = _none_dims(img=g["B"], y=iy, x=ix); out_pels = _none_dims(img=g["B"], y=iy // st, x=ix // st)
        a.set_dims("in_pels", in_pels); a.set_dims("out_pels", out_pels)
        gb = GbtTile(tune.MNt, tune.MNb[0] * tune.MNb[1], (out_pels.dims_prod(), g["OC"]))
        (m_per, n_per), (tm, tn), (bm, bn) = gb.mn_per_thr, gb.thr_per_blk, gb.num_blk
        if m_per % vw or n_per % vw:
            raise UnsupErr("conv_simd only supports work.pels and work.out_chan being multiples of vw")
        pels_pad, oc_pad = bm * tm * m_per, bn * tn * n_per
        fy, fx = max(g["KH"] - st, g["PY"]), max(g["KW"] - st, g["PX"])      # pels the last output position may hang off the last image
        in_pels_pad = _cdiv(in_pels.dims_prod() + fy * in_pels.dstride("y") + fx * in_pels.dstride("x"), vw) * vw
        a.set_dims("work", _none_dims(pels_blk=bm, out_chan_blk=bn, pels_tile=tm, out_chan_tile=tn, pels=m_per, out_chan=n_per))
        a.reset_dims("in", Dims(("chan", "pel"), (g["C"], in_pels_pad), ni.tn))
        a.reset_dims("filts", Dims(("in_chan", "y", "x", "out_chan"), (g["C"], g["KH"], g["KW"], oc_pad), filts.tn))
        a.reset_dims("out", Dims(("chan", "pel"), (oc_pad, pels_pad), no.tn))
        return a
    if fn == "k1conv_simd":      # vector loads / stores, no local memory (src/cnn_op.cc:226-245): in, filts AND out transposed to (chan, pel) forms padded to the blocking
        vw = tune.vw
        if vw not in (2, 4):
            raise UnsupErr(f"CUCL compatibility mode: vector width vw={vw}: HIP has float2 / float4 (the reference's default 8 is an OpenCL type)")
        if m_per % vw or n_per % vw:
            raise UnsupErr("k1conv_simd only supports work.pels and work.out_chan being multiples of vw")
        a.set_u32("vw", vw); a.set_u32("Kb", tune.Kb)
        pels_pad, oc_pad = bm * tm * m_per, bn * tn * n_per
        a.set_dims("work", work)
        a.reset_dims("in", Dims(("chan", "pel"), (g["C"], pels_pad), ni.tn))
        a.reset_dims("filts", Dims(("in_chan", "y", "x", "out_chan"), (g["C"], g["KH"], g["KW"], oc_pad), filts.tn))
        a.reset_dims("out", Dims(("chan", "pel"), (oc_pad, pels_pad), no.tn))
        return a
    if fn == "ipconv":      # inner-product case (one output per channel and image): the reduction is tiled too, over fioc_tile lanes (src/cnn_op.cc:204-209)
        fioc_tile = 4
        while fioc_tile < 32 and fioc_tile * 2 * tm * tn <= 512:
            fioc_tile *= 2
        if g["C"] % fioc_tile:
            raise RtErr(f"ipconv: in_chan={g['C']} must be a multiple of fioc_tile={fioc_tile}")
        work = Dims(work.names + ("fioc_tile",), work.sizes + (fioc_tile,), "none")
    a.set_dims("work", work)
    a.reset_dims("in", in_dims)
    if fn != "ipconv":      # (ipconv reads filts -- and in -- in the reference layout: no layout pass, src/cnn_op.cc:305-311)
        a.reset_dims("filts", Dims(("out_chan_blk", "in_chan", "y", "x", "out_chan_reg", "out_chan_tile"), (bn, g["C"], g["KH"], g["KW"], n_per, tn), filts.tn))
    return a


# ---------------------------------------------------------------------------------------------------------------------------------------
# custom code generation: src/cnn_codegen.cc
# ---------------------------------------------------------------------------------------------------------------------------------------
def _guarded(cg: CallGen, sec: str, n_items: int, body: Callable[[int, str], str], bound: str) -> None:
    """One statement per tpb-sized slice of `n_items` items; the last slice is wrapped in `if( ix < bound ) { ... }` when it overhangs
    (the load pattern every variant uses to fill local memory with all threads)."""
    for i in range(_cdiv(n_items, cg.tpb)):
        ix = f"(LOC_ID_1D + %(tpb) * {i})"
        stmt = body(i, ix)
        if (i + 1) * cg.tpb > n_items:
            cg.line(sec, f"if( {ix} < {bound} ) {{")
            stmt += "}"
        cg.line(sec, stmt)


def _bias_relu(cg: CallGen, work: Dims, tx: int, ty: int) -> str:
    v = f"(out_tile[{ty * work.dsz('out_chan') + tx}] + filts_strip[{tx}])"
    return f"max(0.0f,{v})" if cg.op.get_u32("conv_has_relu") else v


def _filts_and_biases_to_smem(cg: CallGen, filts_smem_sz: int) -> None:
    """Shared by conv / k1conv / tconv (src/cnn_codegen.cc:137-163): contiguous filter slice into filts_smem; later the block's biases, in the
    register-tile order of the transposed filters (out_chan = blk base + (i % out_chan_tile) * out_chan_reg + i / out_chan_tile)."""
    _guarded(cg, "filts_smem_loads", filts_smem_sz, lambda i, ix: f"filts_smem[{ix}] = filts[filts_off+(%(tpb)*{i})];", "%(filts_smem_sz)")
    per_blk = cg.get_arg_dims_by_name("filts").dstride("x")       # out chans per block (== out_chan_tile * out_chan_reg)
    cg.set("out_chan_bias_smem_load_iter", str(_cdiv(per_blk, cg.tpb)))
    cg.line("biases_smem_loads", "int32_t ocix; int32_t const ocix_base = %(GRP_ID_1D_out_chan_blk)*%(filts_x_stride);")
    for i in range(_cdiv(per_blk, cg.tpb)):
        ix = f"(LOC_ID_1D + %(tpb) * {i})"
        cg.line("biases_smem_loads", f"ocix = ocix_base + ({ix} %% %(work_out_chan_tile_dim))*%(work_out_chan_dim) + ( {ix} / %(work_out_chan_tile_dim) );")
        tail = ""
        if (i + 1) * cg.tpb > per_blk:
            cg.line("biases_smem_loads", f"if( {ix} < %(filts_x_stride) ) {{"); tail = "}"
        cg.line("biases_smem_loads", f"if( ocix < %(biases_out_chan_dim) ) {{filts_smem[{ix}] = biases[ocix];}}{tail}")


def _fma_tile(cg: CallGen, sec: str, work: Dims, in_ix: Callable[[int], int]) -> None:
    oc = work.dsz("out_chan")
    for ty in range(work.dsz("pels")):
        for tx in range(oc):
            cg.line(sec, f"out_tile[{ty * oc + tx}] += filts_strip[{tx}]*in_strip[{in_ix(ty)}];")


def gen_sgemm(cg: CallGen) -> None:
    """src/cnn_codegen.cc:409-515: a / b slices of Kb rows into local memory, then per k row: Mt + Nt register loads and Mt*Nt FMAs."""
    work = cg.get_arg_dims_by_name("work")
    Kb, Mt, Nt = work.dsz("Kb"), work.dsz("Mt"), work.dsz("Nt")
    blk = {"a": work.dsz("Mb") * Mt, "b": work.dsz("Nb") * Nt}
    for vn in ("a", "b"):
        d = cg.get_arg_dims_by_name(vn)
        if d.tn != "float":
            raise UnsupErr("CUCL compatibility mode: sgemm on float tensors only")
        row_len, sm_sz, stride = blk[vn], blk[vn] * Kb, d.dstride("K")
        if stride < row_len:
            raise RtErr("sgemm: a block row is longer than the matrix row")
        pad = stride - row_len
        cg.set(f"{vn}_sm_sz", str(sm_sz))
        for i in range(_cdiv(sm_sz, cg.tpb)):
            so = cg.tpb * i
            row, row_off = so // row_len, so % row_len
            extra = f"+(LOC_ID_1D+{row_off})/{row_len}*{pad}" if (pad and row_off + cg.tpb > row_len) else ""   # a slice that crosses block rows
            tail = ""
            if so + cg.tpb > sm_sz:
                cg.line("sm_loads", f"if( (LOC_ID_1D+{so}) < {sm_sz} ) {{"); tail = "}"
            cg.line("sm_loads", f"{vn}_sm[LOC_ID_1D+{so}] = {vn}[{vn}_off+{so + row * pad}{extra}];{tail}")
    for k in range(Kb):
        for m in range(Mt):
            cg.line("inner_loop_body", f"a_r[{m}] = a_sm_off[{m + k * blk['a']}];")
        for n in range(Nt):
            cg.line("inner_loop_body", f"b_r[{n}] = b_sm_off[{n + k * blk['b']}];")
        for m in range(Mt):
            for n in range(Nt):
                cg.line("inner_loop_body", f"c_r[{m * Nt + n}] += a_r[{m}]*b_r[{n}];")
    cg.line("outs_to_b_r", "switch(Mt) { ")
    for m in range(Mt):
        cg.line("outs_to_b_r", f"case {m}:")
        for n in range(Nt):
            cg.line("outs_to_b_r", f"b_r[{n}] = c_r[{m * Nt + n}];")
        cg.line("outs_to_b_r", "break;")
    cg.line("outs_to_b_r", "} ")
    for n in range(Nt):
        cg.line("stores", f"c[c_off+{n}] = b_r[{n}];")


def _sgemm_rows_to_c(cg: CallGen, work: Dims, b_r: Callable[[int], str], store: Callable[[int], str], n_stores: int) -> None:
    """The store pattern all sgemm variants share: a run-time loop over the thread's Mt rows copies row Mt of the register tile into b_r, then stores it."""
    Mt, Nt = work.dsz("Mt"), work.dsz("Nt")
    cg.line("outs_to_b_r", "switch(Mt) { ")
    for m in range(Mt):
        cg.line("outs_to_b_r", f"case {m}:")
        for n in range(Nt):
            cg.line("outs_to_b_r", f"{b_r(n)} = c_r[{m * Nt + n}];")
        cg.line("outs_to_b_r", "break;")
    cg.line("outs_to_b_r", "} ")
    for n in range(n_stores):
        cg.line("stores", store(n))


def _vec_elem(vw: int, ix: int) -> str:
    """Element ix of an array of vw-wide vectors (gva, src/cnn_codegen.cc:284-290): .xyzw up to 4 lanes, .s0 .. .sf beyond (OpenCL only)."""
    if vw > 16:
        raise RtErr("vector width > 16")
    return f"[{ix // vw}].{'xyzw'[ix % vw] if vw <= 4 else 's' + '0123456789abcdef'[ix % vw]}"


def _sgemm_vw(cg: CallGen, work: Dims) -> int:
    vw = cg.op.get_u32("vw")
    if vw not in (2, 4):
        raise UnsupErr(f"CUCL compatibility mode: vector width vw={vw}: HIP has float2 / float4 (the reference's default 8 is an OpenCL type)")
    if work.dsz("Mt") % vw or work.dsz("Nt") % vw:
        raise RtErr("sgemm_simd: Mt and Nt must be multiples of vw")
    return vw


def gen_sgemm_no_local(cg: CallGen) -> None:
    """src/cnn_codegen.cc:387-406 (use_local_mem=0): every thread reads its own Mt + Nt operands of each k row straight from global memory."""
    work = cg.get_arg_dims_by_name("work")
    Kb, Mt, Nt = work.dsz("Kb"), work.dsz("Mt"), work.dsz("Nt")
    ks = {vn: cg.get_arg_dims_by_name(vn).dstride("K") for vn in ("a", "b")}
    for k in range(Kb):
        for m in range(Mt):
            cg.line("inner_loop_body", f"a_r[{m}] = a[a_off+{m + k * ks['a']}];")
        for n in range(Nt):
            cg.line("inner_loop_body", f"b_r[{n}] = b[b_off+{n + k * ks['b']}];")
        for m in range(Mt):
            for n in range(Nt):
                cg.line("inner_loop_body", f"c_r[{m * Nt + n}] += a_r[{m}]*b_r[{n}];")
    _sgemm_rows_to_c(cg, work, lambda n: f"b_r[{n}]", lambda n: f"c[c_off+{n}] = b_r[{n}];", Nt)


def _sgemm_simd_body(cg: CallGen, work: Dims, vw: int, a_src: Callable[[int, int], str], b_src: Callable[[int, int], str]) -> None:
    Kb, Mt, Nt = work.dsz("Kb"), work.dsz("Mt"), work.dsz("Nt")
    for k in range(Kb):
        for m in range(Mt // vw):
            cg.line("inner_loop_body", f"a_r[{m}] = {a_src(m, k)};")
        for n in range(Nt // vw):
            cg.line("inner_loop_body", f"b_r[{n}] = {b_src(n, k)};")
        for m in range(Mt):
            for n in range(Nt):
                cg.line("inner_loop_body", f"c_r[{m * Nt + n}] += a_r{_vec_elem(vw, m)}*b_r{_vec_elem(vw, n)};")
    _sgemm_rows_to_c(cg, work, lambda n: f"b_r{_vec_elem(vw, n)}", lambda n: f"((GASQ float{vw} *)c)[c_off+{n}] = b_r[{n}];", Nt // vw)


def gen_sgemm_simd(cg: CallGen) -> None:
    """src/cnn_codegen.cc:343-385 (use_local_mem=2): as sgemm_no_local with vw-wide vector loads / stores."""
    work = cg.get_arg_dims_by_name("work"); vw = _sgemm_vw(cg, work)
    ks = {}
    for vn in ("a", "b"):
        st = cg.get_arg_dims_by_name(vn).dstride("K")
        if st % vw:
            raise RtErr(f"sgemm_simd: the K stride of {vn} must be a multiple of vw")
        ks[vn] = st // vw
    _sgemm_simd_body(cg, work, vw, lambda m, k: f"((GASQ float{vw} const *)a)[a_off+{m + k * ks['a']}]", lambda n, k: f"((GASQ float{vw} const *)b)[b_off+{n + k * ks['b']}]")


def gen_sgemm_simd_local(cg: CallGen) -> None:
    """src/cnn_codegen.cc:293-341,408-459 (use_local_mem=3): Kb rows of a / b into local memory as vw-wide vectors, vector register loads from there."""
    work = cg.get_arg_dims_by_name("work"); vw = _sgemm_vw(cg, work)
    Kb = work.dsz("Kb")
    blk = {"a": work.dsz("Mb") * work.dsz("Mt") // vw, "b": work.dsz("Nb") * work.dsz("Nt") // vw}
    for vn in ("a", "b"):
        d = cg.get_arg_dims_by_name(vn)
        if d.tn != "float":
            raise UnsupErr("CUCL compatibility mode: sgemm on float tensors only")
        if d.dstride("K") % vw:
            raise RtErr(f"sgemm_simd_local: the K stride of {vn} must be a multiple of vw")
        row_len, sm_sz, stride = blk[vn], blk[vn] * Kb, d.dstride("K") // vw
        if stride < row_len:
            raise RtErr("sgemm: a block row is longer than the matrix row")
        pad = stride - row_len
        cg.set(f"{vn}_sm_sz", str(sm_sz))
        for i in range(_cdiv(sm_sz, cg.tpb)):
            so = cg.tpb * i
            row, row_off = so // row_len, so % row_len
            extra = f"+(LOC_ID_1D+{row_off})/{row_len}*{pad}" if (pad and row_off + cg.tpb > row_len) else ""
            tail = ""
            if so + cg.tpb > sm_sz:
                cg.line("sm_loads", f"if( (LOC_ID_1D+{so}) < {sm_sz} ) {{"); tail = "}"
            cg.line("sm_loads", f"{vn}_sm[LOC_ID_1D+{so}] = ((GASQ %(a_tn){vw} const *)({vn}))[{vn}_off+{so + row * pad}{extra}];{tail}")
    _sgemm_simd_body(cg, work, vw, lambda m, k: f"a_sm_off[{m + k * blk['a']}]", lambda n, k: f"b_sm_off[{n + k * blk['b']}]")


def gen_conv(cg: CallGen) -> None:
    """src/cnn_codegen.cc:165-215: the general variant -- one (in_chan, ky, kx) element per iteration, pels gathered through local memory."""
    work, filts = cg.get_arg_dims_by_name("work"), cg.get_arg_dims_by_name("filts")
    P, OC = work.dsz("pels"), work.dsz("out_chan")
    cg.set("filts_smem_sz", str(filts.dstride("x")))
    _filts_and_biases_to_smem(cg, filts.dstride("x"))
    cg.set("pel_smem_load_iter", str(_cdiv(P * work.dsz("pels_tile"), cg.tpb)))
    cg.set("out_chan_tile", "(%(LOC_ID_1D_out_chan_tile)+%(GRP_ID_1D_out_chan_blk)*%(work_out_chan_tile_dim))")
    cg.set("pel_tile", "(%(LOC_ID_1D_pels_tile)+%(GRP_ID_1D_pels_blk)*%(work_pels_tile_dim))")
    cg.set("out_chan_ix", "(%(out_chan_tile)*%(work_out_chan_dim))")
    for i in range(P):
        cg.insert_nda_ix_exprs(f"pel_ix_{i}", cg.all_ix_dims["out_pel_ix"], f"(%(pel_tile)*%(work_pels_dim)+{i})")
    for tx in range(OC):
        cg.line("loads", f"filts_strip[{tx}] = filts_smem[%(LOC_ID_1D_out_chan_tile)+{tx}*%(work_out_chan_tile_dim)];")
    for ty in range(P):
        cg.line("loads", f"in_strip[{ty}] = in_smem[%(LOC_ID_1D_pels_tile)*%(work_pels_dim)+{ty}];")
    cg.line("stores", "int32_t tpix[%(work_pels_dim)];")
    cg.line("stores", "int32_t tcix[%(work_out_chan_dim)];")
    for ty in range(P):
        cg.line("stores", f"tpix[{ty}] = %(pel_ix_{ty}_img)*%(out_img_stride) + ( %(pel_ix_{ty}_x_nomod) %% (%(out_y_dim)*%(out_x_dim)) );")
    for tx in range(OC):
        cg.line("stores", f"  tcix[{tx}] = (%(out_chan_ix)+{tx})*%(out_chan_stride);")
    _fma_tile(cg, "fmas", work, lambda ty: ty)
    for ty in range(P):
        cg.line("stores", f"if( %(pel_ix_{ty}_x_nomod) >= %(pel_ix_0_dims_prod) ) {{ return; }}")
        for tx in range(OC):
            cg.line("stores", f"if( tcix[{tx}] < (%(out_chan_dim)*%(out_chan_stride)) ) {{ out[ tpix[{ty}] + tcix[{tx}] ] = {_bias_relu(cg, work, tx, ty)}; }}")


def gen_k1conv(cg: CallGen) -> None:
    """src/cnn_codegen.cc:625-761: 1x1 / stride 1 / no padding; in is blk:blk_iter:blk_iter_chan:blk_pel; out in the reference layout, or -- write-xposed -- in the
    next k1conv's input layout."""
    work, filts, inp, out = (cg.get_arg_dims_by_name(n) for n in ("work", "filts", "in", "out"))
    st, pad = cg.get_arg_dims_by_name("stride"), cg.get_arg_dims_by_name("in_pad")
    if pad.sizes != (0, 0) or st.sizes != (1, 1) or filts.dsz("x") != 1 or filts.dsz("y") != 1:
        raise RtErr("k1conv needs a 1x1 kernel, stride 1 and no padding")
    P, OC = work.dsz("pels"), work.dsz("out_chan")
    filts_smem_sz = filts.dstride("in_chan") * inp.dsz("blk_iter_chan")
    out_smem_sz = work.dsz("pels_tile") * work.dsz("out_chan_tile") * P
    cg.set("filts_smem_sz", str(filts_smem_sz)); cg.set("out_smem_sz", str(out_smem_sz))
    cg.set("all_smem_sz", str(max(out_smem_sz, filts_smem_sz + inp.dstride("blk_iter"))))
    _filts_and_biases_to_smem(cg, filts_smem_sz)
    _guarded(cg, "smem_loads", inp.dstride("blk_iter"), lambda i, ix: f"    in_smem[{ix}] = in[ blk_in_ix_base + (%(tpb)*{i}) ];", "%(in_blk_iter_stride)")
    cg.set("out_chan_tile", "(%(GRP_ID_1D_out_chan_blk)*%(work_out_chan_tile_dim)+%(LOC_ID_1D_out_chan_tile))")
    cg.set("out_chan_ix", "(%(out_chan_tile)*%(work_out_chan_dim))")
    if out.has("blk"):
        # write-xposed chaining (src/cnn_codegen.cc:656-707, src/rtc_fwd.cc:495-503): `out` has the NEXT k1conv's input dims (blk:blk_iter:blk_iter_chan:blk_pel), so
        # that layer needs no k1conv_xpose_in pass.  Per per-thread out_chan the block's values go through local memory and leave as (mostly) sequential runs of pels.
        if work.dsz("out_chan_blk") * work.dsz("out_chan_tile") * OC != out.dsz("blk_iter") * out.dsz("blk_iter_chan"):
            raise RtErr("k1conv write-xposed: padded out_chans of this layer != padded in_chans of the next")
        if work.dsz("pels_blk") * work.dsz("pels_tile") * P != out.dsz("blk") * out.dsz("blk_pel") or out.dsz("blk_pel") != inp.dsz("blk_pel"):
            raise RtErr("k1conv write-xposed: the two layers' pel blockings differ")
        cg.line("stores", "int32_t const out_ix = (%(GRP_ID_1D_out_chan_blk)*%(work_out_chan_tile_dim)*%(work_out_chan_dim))*%(out_blk_iter_chan_stride) + %(GRP_ID_1D_pels_blk)*%(out_blk_stride);")
        cg.line("stores", "int32_t xpbuf_rd_pel;")
        cg.line("stores", "int32_t xpbuf_rd_chan;")
        for tx in range(OC):
            cg.line("stores", "  BARRIER_SYNC;")
            for ty in range(P):
                cg.line("stores", f"out_smem_off[%(tpb)*{ty}] = {_bias_relu(cg, work, tx, ty)};")
            cg.line("stores", "  BARRIER_SYNC;")
            for ty in range(P):
                obe = f"(LOC_ID_1D + %(tpb)*{ty})"
                cg.line("stores", f"  xpbuf_rd_pel = {obe} %% %(out_blk_pel_dim) ;")
                cg.line("stores", f"  xpbuf_rd_chan = {obe} / %(out_blk_pel_dim) ;")
                cg.line("stores", f"out[out_ix + xpbuf_rd_pel + (xpbuf_rd_chan*%(work_out_chan_dim)+{tx})*%(out_blk_iter_chan_stride)] = "
                                  "all_smem[xpbuf_rd_chan+(xpbuf_rd_pel %% %(work_pels_dim))*%(tpb)+ (xpbuf_rd_pel / %(work_pels_dim))*%(work_out_chan_tile_dim) ];")
    else:
        _k1conv_ref_stores(cg, work, P, OC)
    for ty in range(P):
        for tx in range(OC):
            cg.line("dummy_stores", f"out_off[{(ty * OC + tx) * cg.tpb}] = {_bias_relu(cg, work, tx, ty)};")
    for tx in range(OC):
        cg.line("bias_loads", f"filts_strip[{tx}] = filts_smem_off[{tx}*%(work_out_chan_tile_dim)];")
    if inp.dsz("blk_pel") != work.dsz("pels_tile") * P:
        raise RtErr("k1conv: in.blk_pel != pels_tile * pels")
    for ic in range(inp.dsz("blk_iter_chan")):
        for tx in range(OC):
            cg.line("inner_loop_body", f"filts_strip[{tx}] = filts_smem_off[({ic}*%(filts_in_chan_stride))+{tx}*%(work_out_chan_tile_dim)];")
        for ty in range(P):
            cg.line("inner_loop_body", f"in_strip[{ty}] = in_smem_off[({ic}*%(in_blk_pel_dim)+{ty})];")
        _fma_tile(cg, "inner_loop_body", work, lambda ty: ty)


def _k1conv_ref_stores(cg: CallGen, work: Dims, P: int, OC: int) -> None:
    """k1conv's stores into the reference output layout (src/cnn_codegen.cc:708-731)."""
    cg.line("stores", "  int32_t tpix[%(work_pels_dim)];")
    cg.line("stores", "  int32_t tcix[%(work_out_chan_dim)];")
    for ty in range(P):
        cg.insert_nda_ix_exprs(f"out_pel_{ty}", cg.all_ix_dims["out_ref_pel"],
                               f"( (%(GRP_ID_1D_pels_blk)*%(work_pels_tile_dim) + %(LOC_ID_1D_pels_tile))*%(work_pels_dim) + {ty} )")
        cg.line("stores", f"  tpix[{ty}] = %(out_pel_{ty}_img)*%(out_img_stride) +  %(out_pel_{ty}_x)*%(out_x_stride) + %(out_pel_{ty}_y)*%(out_y_stride);")
    for tx in range(OC):
        cg.line("stores", f"  tcix[{tx}] = (%(out_chan_ix)+{tx})*%(out_chan_stride);")
    for ty in range(P):
        cg.line("stores", f"  if( %(out_pel_{ty}_img) >= %(out_img_dim) ) {{ return; }}")
        for tx in range(OC):
            cg.line("stores", f"if( tcix[{tx}] < (%(out_chan_dim)*%(out_chan_stride)) ) {{ out[ tpix[{ty}] + tcix[{tx}] ] = {_bias_relu(cg, work, tx, ty)}; }}")


def gen_tconv(cg: CallGen) -> None:
    """src/cnn_codegen.cc:763-823: a block handles blk_y output lines x `pels` columns; the input tile (expanded copy, tconv_xpose_in) is
    walked one input channel and one kernel row at a time, unrolled over kx."""
    stride, work, filts, inp = (cg.get_arg_dims_by_name(n) for n in ("stride", "work", "filts", "in"))
    P, OC = work.dsz("pels"), work.dsz("out_chan")
    cg.set("filts_smem_sz", str(filts.dstride("y")))
    _filts_and_biases_to_smem(cg, filts.dstride("y"))
    cg.line("filts_smem_loads", "filts_off += %(filts_smem_sz);")
    _guarded(cg, "in_smem_loads", inp.dstride("blk_in_chan"), lambda i, ix: f"in_smem[{ix}] = in[ blk_in_ix_base + (%(tpb)*{i}) ];", "%(in_blk_in_chan_stride)")
    cg.line("in_smem_loads", "blk_in_ix_base += %(in_blk_in_chan_stride);")
    for i in range(inp.dsz("blk_x")):
        cg.line("inner_loop_body", f"in_strip[{i}] = in_smem_off[{i}];")
    if work.dsz("out_chan_tile") != filts.dsz("out_chan_tile"):
        raise RtErr("tconv: work / filts out_chan_tile mismatch")
    for kx in range(filts.dsz("x")):
        for tx in range(OC):
            cg.line("inner_loop_body", f"filts_strip[{tx}] = filts_smem_off[{kx}*%(filts_x_stride)+{tx}*%(filts_out_chan_reg_stride)];")
        _fma_tile(cg, "inner_loop_body", work, lambda ty: ty * stride.dsz("x") + kx)
    for tx in range(OC):
        cg.line("bias_loads", f"filts_strip[{tx}] = filts_smem_off[{tx}*%(filts_out_chan_reg_stride)];")
    cg.line("stores", "if( %(out_line_img) >= %(out_img_dim) ) { return; }")
    cg.line("stores", "int32_t out_x = %(GRP_ID_1D_blk_bx)*%(work_pels_dim);")
    cg.line("stores", "int32_t out_chan = (%(GRP_ID_1D_out_chan_blk)*%(work_out_chan_tile_dim) + %(LOC_ID_1D_out_chan_tile))*%(work_out_chan_dim);")
    cg.line("stores", "GASQ float * out_off = out + %(out_line_img)*%(out_img_stride) + out_chan*%(out_chan_stride) + %(out_line_y)*%(out_y_stride) + out_x*%(out_x_stride) ;")
    for ty in range(P):
        cg.line("stores", f"if( (out_x + {ty}) >= %(out_x_dim) ) {{ return; }} // this x value and the following are off-the-end pels, so don't store them.")
        for tx in range(OC):
            cg.line("stores", f"if( (out_chan + {tx}) < %(out_chan_dim) ) {{ out_off[ {tx}*%(out_chan_stride) + {ty}*%(out_x_stride) ] = {_bias_relu(cg, work, tx, ty)}; }}")


def gen_ipconv(cg: CallGen) -> None:
    """src/cnn_codegen.cc:217-283: the inner-product variant (output 1x1, no padding: AlexNet fc6-fc8).  in and filts are read in the reference
    layout, where an image / a filter is one contiguous vector of in_chan*y*x elements; fioc_tile consecutive lanes share one (pels, out_chan)
    register tile and each takes every fioc_tile-th element of the reduction; the partial sums meet through __shfl_down over groups of
    fioc_tile lanes (a power of two <= 32: such groups never straddle a 64-wide wavefront), lane 0 of a group stores."""
    work = cg.get_arg_dims_by_name("work")
    P, OC, F = work.dsz("pels"), work.dsz("out_chan"), work.dsz("fioc_tile")
    for what, arr, row_sz, row_stride in (("filts", "filts", work.dsz("out_chan_tile") * OC * F, "%(filts_out_chan_stride)"),
                                          ("in", "in", work.dsz("pels_tile") * P * F, "%(in_img_stride)")):
        cg.set(f"{what}_smem_sz", str(row_sz))
        # element ix of the local buffer = (row ix / fioc_tile, lane % fioc_tile): row = ix / fioc_tile of the block's filters / images
        _guarded(cg, f"{what}_smem_loads", row_sz,
                 lambda i, ix, arr=arr, what=what, row_stride=row_stride:
                 f"{what}_smem[{ix}] = {arr}[{what}_off+(( LOC_ID_1D/%(work_fioc_tile_dim) + %(tpb)/%(work_fioc_tile_dim)* {i})*{row_stride})];",
                 f"%({what}_smem_sz)")
    for tx in range(OC):
        cg.line("loads", f"filts_strip[{tx}] = filts_smem_off[{tx}*%(work_fioc_tile_dim)];")
    for ty in range(P):
        cg.line("loads", f"in_strip[{ty}] = in_smem_off[{ty}*%(work_fioc_tile_dim)];")
    _fma_tile(cg, "fmas", work, lambda ty: ty)
    # the store loop runs over the thread's pels at run time: row work_pel of the register tile is copied into filts_strip first
    cg.line("outs_to_filts_strip", "if( (in_pel+work_pel) >= %(in_img_dim) ) { return; }")
    cg.line("outs_to_filts_strip", "switch(work_pel) { ")
    for ty in range(P):
        cg.line("outs_to_filts_strip", f"case {ty}:")
        for tx in range(OC):
            cg.line("outs_to_filts_strip", f"filts_strip[{tx}] = out_tile[{ty * OC + tx}];")
        cg.line("outs_to_filts_strip", "break;")
    cg.line("outs_to_filts_strip", "} ")
    for tx in range(OC):
        wb = F // 2
        while wb:
            cg.line("stores", f"filts_strip[{tx}] += __shfl_down( filts_strip[{tx}], {wb}, {F} );"); wb //= 2
        v = f"(filts_strip[{tx}] + biases[ocix+{tx}])"
        v = f"max(0.0f,{v})" if cg.op.get_u32("conv_has_relu") else v
        cg.line("stores", f"if( (%(LOC_ID_1D_fioc_tile) == 0 ) && ((ocix + {tx}) < %(out_chan_dim)) ) {{ out[out_off + {tx}*%(out_chan_stride)] = {v}; }}")


def gen_k1conv_simd(cg: CallGen) -> None:
    """src/cnn_codegen.cc:514-562 (k1conv with use_local_mem=2): no local memory; per in_chan a thread reads pels/vw + out_chan/vw vectors of the
    (chan, pel) input and the (in_chan, ..., out_chan) filters, Kb in_chans per loop trip; biases arrive in filts_strip, ReLU in the store rows."""
    work = cg.get_arg_dims_by_name("work")
    vw = cg.op.get_u32("vw"); Kb = cg.op.get_u32("Kb")
    P, OC = work.dsz("pels"), work.dsz("out_chan")
    if P % vw or OC % vw:
        raise RtErr("k1conv_simd: work.pels and work.out_chan must be multiples of vw")
    ics, fics = cg.get_arg_dims_by_name("in").dstride("chan"), cg.get_arg_dims_by_name("filts").dstride("in_chan")
    if ics % vw or fics % vw:
        raise RtErr("k1conv_simd: the chan strides of in / filts must be multiples of vw")
    ics //= vw; fics //= vw
    for k in range(Kb):
        for tx in range(P // vw):
            cg.line("inner_loop_body", f"in_strip[{tx}] = ((GASQ float{vw} const *)in)[in_off+{tx + k * ics}];")
        for ty in range(OC // vw):
            cg.line("inner_loop_body", f"filts_strip[{ty}] = ((GASQ float{vw} const *)filts)[filts_off+{ty + k * fics}];")
        for tx in range(P):
            for ty in range(OC):
                cg.line("inner_loop_body", f"out_tile[{tx * OC + ty}] += in_strip{_vec_elem(vw, tx)}*filts_strip{_vec_elem(vw, ty)};")
    relu = cg.op.get_u32("conv_has_relu")
    cg.line("outs_to_in_strip", "switch(ty) { ")
    for ty in range(OC):
        cg.line("outs_to_in_strip", f"case {ty}:")
        for tx in range(P):
            v = f"(out_tile[{tx * OC + ty}]+filts_strip{_vec_elem(vw, ty)})"
            cg.line("outs_to_in_strip", f"in_strip{_vec_elem(vw, tx)} = {('max(0.0f,' + v + ')') if relu else v};")
        cg.line("outs_to_in_strip", "break;")
    cg.line("outs_to_in_strip", "} ")
    for tx in range(P // vw):
        cg.line("stores", f"((GASQ float{vw} *)out)[out_off+{tx}] = in_strip[{tx}];")


def gen_conv_simd(cg: CallGen) -> None:
    """src/cnn_codegen.cc:564-623 (the general conv with use_local_mem=2): one (in_chan, ky, kx) element per loop trip; the thread's pels are consecutive
    positions of the padded output grid, their inputs stride_x apart (plus a row jump where the run wraps, for stride > 1); filters as vectors."""
    work, in_pels, stride = cg.get_arg_dims_by_name("work"), cg.get_arg_dims_by_name("in_pels"), cg.get_arg_dims_by_name("stride")
    vw = cg.op.get_u32("vw")
    P, OC = work.dsz("pels"), work.dsz("out_chan")
    if P % vw or OC % vw:
        raise RtErr("conv_simd: work.pels and work.out_chan must be multiples of vw")
    if cg.op.get_u32("Kb") != 1 or stride.dsz("y") != stride.dsz("x"):
        raise RtErr("conv_simd: Kb == 1 and a uniform stride are required")
    fics = cg.get_arg_dims_by_name("filts").dstride("in_chan")
    if fics % vw:
        raise RtErr("conv_simd: the in_chan stride of filts must be a multiple of vw")
    row_extra = (stride.dsz("y") - 1) * in_pels.dstride("y")
    for tx in range(P):
        reo = f"+((out_x + {tx})/%(out_pels_x_dim)*{row_extra})" if row_extra else ""
        cg.line("inner_loop_body", f"in_strip{_vec_elem(vw, tx)} = in[in_off+{tx * stride.dsz('x')} {reo}];")
    for ty in range(OC // vw):
        cg.line("inner_loop_body", f"filts_strip[{ty}] = ((GASQ float{vw} const *)filts)[filts_off+{ty}];")
    for tx in range(P):
        for ty in range(OC):
            cg.line("inner_loop_body", f"out_tile[{tx * OC + ty}] += in_strip{_vec_elem(vw, tx)}*filts_strip{_vec_elem(vw, ty)};")
    relu = cg.op.get_u32("conv_has_relu")
    cg.line("outs_to_in_strip", "switch(ty) { ")
    for ty in range(OC):
        cg.line("outs_to_in_strip", f"case {ty}:")
        for tx in range(P):
            v = f"(out_tile[{tx * OC + ty}]+filts_strip{_vec_elem(vw, ty)})"
            cg.line("outs_to_in_strip", f"in_strip{_vec_elem(vw, tx)} = {('max(0.0f,' + v + ')') if relu else v};")
        cg.line("outs_to_in_strip", "break;")
    cg.line("outs_to_in_strip", "} ")
    for tx in range(P // vw):
        cg.line("stores", f"((GASQ float{vw} *)out)[out_off+{tx}] = in_strip[{tx}];")


_EMITTERS: Dict[str, Callable[[CallGen], None]] = {"k1conv_simd": gen_k1conv_simd, "conv_simd": gen_conv_simd, "sgemm": gen_sgemm, "sgemm_no_local": gen_sgemm_no_local, "sgemm_simd": gen_sgemm_simd,
                                                   "sgemm_simd_local": gen_sgemm_simd_local, "conv": gen_conv, "k1conv": gen_k1conv, "tconv": gen_tconv, "ipconv": gen_ipconv}


def custom_codegen(cg: CallGen, template_name: str) -> None:
    """cnn_custom_codegen_t::gen_op (src/cnn_codegen.cc:11-27): templates without a hook pass through untouched."""
    if template_name in _EMITTERS:
        _EMITTERS[template_name](cg)
    elif template_name == "reduce":      # gen_op_reduce (src/cnn_codegen.cc:28-34): one accumulation line per member of the `ins` pack
        for vn in cg.multi_args.get("ins", []):
            cg.line("ins_ops", f"v += {vn}[GLOB_ID_1D];")
    elif template_name in ("bconv", "bconv_fb"):
        raise UnsupErr(f"CUCL compatibility mode: the custom code generation of '{template_name}' is not restated")


def chain_k1conv(first: Op, second: Op) -> None:
    """In place: `first` (an annotated k1conv whose output feeds only `second`, another annotated k1conv) writes its output in `second`'s input layout
    (conv_pipe_fwd_t::init with enable_write_xpose, src/rtc_fwd.cc:495-503): second then runs without its k1conv_xpose_in pass."""
    if first.get_func_name() != "k1conv" or second.get_func_name() != "k1conv":
        raise UnsupErr("write-xposed chaining needs two k1conv functions")
    if first.get_dims("out_ref") != second.get_dims("in_ref"):
        raise RtErr("write-xposed chaining: the second layer does not read the first one's output")
    first.reset_dims("out", second.get_dims("in"))


# ---------------------------------------------------------------------------------------------------------------------------------------
# instantiating the reference's templates
# ---------------------------------------------------------------------------------------------------------------------------------------
def xpose_ops(anno: Op) -> List[Tuple[str, str, str, Op]]:
    """The layout passes an annotated conv needs before its main function: (template name, source arg, destination arg, op for the template).
    filts always (xpose_filts, src/rtc_fwd.cc:229-243, src/rtc_prof.cc:99-101); in for k1conv / tconv (<func>_xpose_in)."""
    fn = anno.get_func_name()
    res: List[Tuple[str, str, str, Op]] = []
    if fn in ("conv", "k1conv", "tconv"):
        res.append(("xpose_filts", "filts_ref", "filts", anno))
    if fn in ("k1conv", "tconv"):
        res.append((fn + "_xpose_in", "in_ref", "in", anno))
    if fn in ("k1conv_simd", "conv_simd"):
        res.append((fn + "_xpose_filts", "filts_ref", "filts", anno)); res.append((fn + "_xpose_in", "in_ref", "in", anno))
    return res


def post_xpose_ops(anno: Op) -> List[Tuple[str, str, str, Op]]:
    """Layout passes AFTER the main function (src/rtc_prof.cc:117-120): variants that write a transposed `out` get <func>_xpose_out (out -> out_ref)."""
    fn = anno.get_func_name()
    return [(fn + "_xpose_out", "out", "out_ref", anno)] if fn in ("k1conv_simd", "conv_simd") else []


def instantiate_ref(rtc_dir: str, template_name: str, anno: Op, gen_fn: str) -> Instance:
    """One generated function from the reference's template `template_name`.cucl (read from a Boda checkout) for the annotated op."""
    t = load_template(rtc_dir, template_name)
    return instantiate(t, anno, gen_fn, custom=custom_codegen)
