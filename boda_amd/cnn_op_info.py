"""`cnn_op_info` mode: per-op info and EFFICIENCY rows for a list of ops (the tables of the reference's write-ups).

Restates cnn_op_info_t::main (src/cnn-prof.cc:24-130) and conv_op_info_to_latex_t (src/latex-util.H:21-140) -- behaviour, not code:
every op line is annotated with --op-tune, profiled with profile_rcg_call on --rtc (inputs generated on the device when --gen-data-mode
is given, zeros otherwise), optionally profiled a second time under --op-tune-comp on --rtc-comp and compared var by var (max-rel-diff
against --mrd-toler, per-var overrides); one row per op goes to the info table (--op-info-tab-fn) and one to the efficiency table
(--op-eff-tab-fn), in the reference's LaTeX-row text:
    Convolution info row   KSZ & Stride & out_chans & B & $ y \\dx x \\dx chan $(in) & $..$(out) & MxKxN & Bytes & FLOPs & F/B \\\\
    Convolution eff row    KSZ & Stride & out_chans & $ B \\dx y \\dx x \\dx chan $ & \\verb|type| & [MxKxN & Bytes & FLOPs & F/B &] Runtime & F/s & %Peak \\\\
    sgemm eff row          MxKxN & Bytes & FLOPs & F/B & Runtime(comp) & F/s(comp) & Runtime & F/s & speedup \\\\
with flops = 2*M*N*K and bytes = 4*(in + out + filts + biases) (sgemm: 4*(a + b + c)) -- src/latex-util.H:116-120,126-133 -- and numbers in
the reference's engineering notation (pp_val, src/str_util.cc:230-256).  The text output ends with ***ALL IS WELL*** / ***MAD FAILS***.
    python -m boda_amd.cnn_op_info --cnn-func-sigs-fn tests/golden/ops/conv-ops-debug.txt --gen-data-mode 5 --op-eff-tab-fn eff.tex
        [--rtc-comp '(be=cpu)'] [--peak-flops 157.3e12] [--print-format 0|1] [--inc-op-info-in-eff 1]
"""
from __future__ import annotations
import argparse
import math
import sys
from typing import Dict, List, Optional, TextIO

from .cnn_op import OpTune, add_codegen_annotations
from .digest import SsdsDiff
from .op import Op, RtErr, UnsupErr, read_ops

PEAK_FP32_MFMA = 157.3e12   # the reference's default is its own GPU's 6600e9 (src/cnn-prof.cc:38); here: MI355X fp32 MFMA


# ---- number formatting: engineering notation with three significant digits and a size suffix (src/str_util.cc:230-256)
def _pp_part(v: float, force: bool) -> str:
    if v < 10.0:
        return f"{v:.2f}"
    if v < 100.0:
        return f"{v:.1f}"
    if v < 1000.0 or force:
        return f"{v:.0f}"
    return "***"


def pp_val(orig: float) -> str:
    if math.isnan(orig):
        return "NAN"
    if orig < 0.0:
        raise RtErr("pp_val: negative value")
    v, exp = float(orig), 0
    while v < 1.0:
        v *= 1000.0; exp -= 1
        if exp < -4:
            return repr(float(orig))
    while True:
        ret = _pp_part(v, False)
        if ret != _pp_part(1e6, exp == 5):
            break
        v /= 1000.0; exp += 1
    if exp < 0:
        return ret + "munp"[-1 - exp]
    if exp == 0:
        return ret
    return ret + "KMGTP"[exp - 1]


pp_secs = lambda v: pp_val(v) + "s"
pp_flops = lambda v: pp_val(v) + "F"
pp_bytes = lambda v: pp_val(v) + "B"
pp_fps = lambda v: pp_val(v) + "F/s"
pp_bps = lambda v: pp_val(v) + "B/s"


def _yxc(d, include_img: bool = False) -> str:
    return "$ %s %s \\dx %s \\dx %s $" % ((str(d.dsz("img")) + " \\dx") if include_img else "", d.dsz("y"), d.dsz("x"), d.dsz("chan"))


def _mkn(M: int, K: int, N: int) -> str:
    return f"$ {M} $" if (M == K == N) else f"$ {M} \\dx {K} \\dx {N} $"


class OpInfoToLatex:
    """One op's rows (conv_op_info_to_latex_t).  print_format 0: engineering notation, 1: raw numbers."""

    def __init__(self, op: Op, print_format: int = 0, inc_op_info_in_eff: int = 0, show_bytes_and_ai: bool = True):
        self.op, self.print_format, self.inc, self.show = op, print_format, inc_op_info_in_eff, show_bytes_and_ai
        self.emit_bw = False
        t = op.get_type()
        if t == "Convolution":
            g = op.conv_geom()
            self.din, self.dout = op.get_dims("in"), op.get_dims("out")
            self.B = g["B"]
            self.M, self.K, self.N = g["B"] * g["OH"] * g["OW"], g["C"] * g["KH"] * g["KW"], g["OC"]
        elif t == "sgemm":
            g = op.sgemm_geom()
            self.B, self.M, self.K, self.N = 1, g["M"], g["K"], g["N"]
        else:
            raise RtErr("cnn-op-info: unhandled op: " + t)
        self.flops, self.bytes = op.flops(), op.algo_bytes()

    def _pp(self, f, v):
        return f(v) if self.print_format == 0 else repr(float(v))

    def base_info(self) -> str:
        if self.op.get_type() != "Convolution":
            return ""
        g = self.op.conv_geom()
        if g["KH"] != g["KW"] or g["SY"] != g["SX"]:
            raise RtErr("cnn-op-info: kernel size and stride must be square")   # (the reference asserts it)
        return f"{g['KH']} & {g['SY']} & {self.dout.dsz('chan')}"

    def ai_mkn(self) -> str:
        if self.show:
            return " %s & %s & %s & %s " % (_mkn(self.M, self.K, self.N), self._pp(pp_bytes, self.bytes), self._pp(pp_flops, self.flops), self._pp(pp_val, self.flops / self.bytes))
        if self.print_format == 2:
            return "%.3g" % float(self.flops)           # (wis-ana's ops table: the one place the reference checks print_format == 2, src/latex-util.H:52-53)
        return " %s " % self._pp(pp_flops, self.flops)

    def info_row(self, brief: bool = False) -> str:
        s = self.base_info()
        if self.op.get_type() == "Convolution":
            s += f" & {self.B} & {_yxc(self.din)} & "
            if not brief:
                s += f"{_yxc(self.dout)} & "
        return s + self.ai_mkn() + "\\\\ \n"

    def eff_row(self, rtc_op_type: str, secs: float, peak_flops: float, secs_comp: float = float("nan")) -> str:
        if self.op.get_type() == "sgemm":
            s = self.ai_mkn()
            s += " & %s & %s " % (self._pp(pp_secs, secs_comp), self._pp(pp_fps, self.flops / secs_comp if secs_comp == secs_comp else float("nan")))
            s += " & %s & %s " % (self._pp(pp_secs, secs), self._pp(pp_fps, self.flops / secs))
            s += " & %.2fx " % (secs_comp / secs)
        else:
            s = self.base_info() + " & %s & \\verb|%s| & " % (_yxc(self.din, True), rtc_op_type)
            if self.inc:
                s += self.ai_mkn() + " & "
            fps = self.flops / secs
            s += " %s & %s & %s " % (self._pp(pp_secs, secs), self._pp(pp_fps, fps), self._pp(pp_val, fps / peak_flops * 100.0))
            if self.emit_bw:
                s += " -- %s %s --" % (self._pp(pp_bps, self.bytes / secs), self._pp(pp_val, self.bytes / secs / 8.0e12 * 100.0))   # (of the 8 TB/s HBM peak)
        return s + "\\\\ \n"


def cnn_op_info(rtc, ops: List[Op], op_tune: OpTune, gen_mode: Optional[int] = None, run_iter: int = 1, rtc_comp=None, op_tune_comp: Optional[OpTune] = None,
                mrd_toler: float = 2e-4, var_mrd_toler: Optional[Dict[str, float]] = None, peak_flops: float = PEAK_FP32_MFMA, print_format: int = 0,
                inc_op_info_in_eff: int = 0, out: TextIO = sys.stdout, info_out: Optional[TextIO] = None, eff_out: Optional[TextIO] = None) -> int:
    """-> number of comparison failures.  Needs initialised backends (the run is the point of this mode)."""
    from .ops_prof import OpsBackend, profile_rcg_call
    be = OpsBackend(rtc)
    be_comp = OpsBackend(rtc_comp) if rtc_comp is not None else None
    num_mad_fail = 0
    for op in ops:
        tl = OpInfoToLatex(op, print_format, inc_op_info_in_eff)
        if info_out is not None:
            info_out.write(tl.info_row(False))
        anno = add_codegen_annotations(op, op_tune)
        vs1, prc = profile_rcg_call(be, anno, gen_mode, 0.0, run_iter, want_outs=be_comp is not None, tile=op_tune.hip_tile)
        secs_comp = float("nan")
        if be_comp is not None:
            anno_c = add_codegen_annotations(op, op_tune_comp or OpTune())
            vs2, prc_c = profile_rcg_call(be_comp, anno_c, gen_mode, 0.0, 1, want_outs=True)
            secs_comp = prc_c.rt_secs
            if sorted(vs1) != sorted(vs2):
                raise RtErr(f"reg/comp out var set mismatch: vns1={sorted(vs1)} vns2={sorted(vs2)}")
            out.write(f"vars_to_compare: {sorted(vs1)}\n")
            for vn in sorted(vs1):
                sd = SsdsDiff.of(vs1[vn], vs2[vn]); tol = (var_mrd_toler or {}).get(vn, mrd_toler)
                if sd.has_nan() or sd.mrd >= tol:
                    num_mad_fail += 1
                    out.write(f"{vn}: {op.get_dims(vn).pretty()} ssds_str(out_batch_1,out_batch_2)={sd.basic_str()}\n")
        if eff_out is not None:
            eff_out.write(tl.eff_row(anno.get_type(), prc.rt_secs, peak_flops, secs_comp))
    rtc.finish_and_sync()
    if rtc_comp is not None:
        rtc_comp.finish_and_sync()
    out.write("***ALL IS WELL***\n" if not num_mad_fail else f"***MAD FAILS*** num_mad_fail={num_mad_fail}\n")
    return num_mad_fail


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="boda_amd.cnn_op_info", description=__doc__.split("\n")[0])
    ap.add_argument("--cnn-func-sigs-fn", required=True)
    ap.add_argument("--out-fn"); ap.add_argument("--op-info-tab-fn"); ap.add_argument("--op-eff-tab-fn")
    ap.add_argument("--print-format", type=int, default=0); ap.add_argument("--inc-op-info-in-eff", type=int, default=0)
    ap.add_argument("--peak-flops", type=float, default=PEAK_FP32_MFMA)
    ap.add_argument("--op-tune", default="()"); ap.add_argument("--op-tune-comp", default="()")
    ap.add_argument("--run-iter", type=int, default=1); ap.add_argument("--gen-data-mode", type=int)
    ap.add_argument("--rtc", default="(be=hip)"); ap.add_argument("--rtc-comp"); ap.add_argument("--mrd-toler", type=float, default=2e-4)
    a = ap.parse_args(argv)
    from .rtc import make_rtc
    rtc = make_rtc(a.rtc); rtc.init()
    rtc_comp = None
    if a.rtc_comp:
        rtc_comp = make_rtc(a.rtc_comp); rtc_comp.init()
    files = [open(fn, "w") if fn else None for fn in (a.out_fn, a.op_info_tab_fn, a.op_eff_tab_fn)]
    try:
        n = cnn_op_info(rtc, read_ops(a.cnn_func_sigs_fn), OpTune.parse(a.op_tune), a.gen_data_mode, a.run_iter, rtc_comp, OpTune.parse(a.op_tune_comp), a.mrd_toler,
                        None, a.peak_flops, a.print_format, a.inc_op_info_in_eff, files[0] or sys.stdout, files[1], files[2])
    finally:
        for f in files:
            if f:
                f.close()
    return 1 if n else 0


if __name__ == "__main__":
    sys.exit(main())
