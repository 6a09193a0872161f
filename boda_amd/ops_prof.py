"""Per-op profile + parity harness: the `ops-prof` / `cnn_op_info` modes over be=hip.

Restates the protocol of the reference's harness (behaviour, not code):
  * profile_rcg_call   src/rtc_prof.cc:44-126   create vars -> generate inputs ON DEVICE -> run run_iter times ->
                                                read outputs back -> finish_and_sync -> get_dur(last call) in secs
  * ops_prof_t::main   src/rtc_prof.cc:194-371  per op x per tune; first (kg) tune = known good; every tune is compared
                                                to it (max-rel-diff, default 2e-4) and to the digests of an input
                                                wisdom file; results optionally written as a wisdom file; the text
                                                output ends with ***ALL IS WELL*** or ***MAD FAILS***
  * efficiency rows    src/latex-util.H:108-134 flops = 2*M*N*K, bytes = 4*(in+out+filts+biases)
CLI:  python -m boda_amd.ops_prof --ops-fn F --op-tunes '(def=(),...)' --kg-tune-tag def [--gen-data-mode 5]
      [--wisdom-in-fn W] [--wisdom-out-fn W] [--run-iter N] [--write-runs 1]
"""
from __future__ import annotations
import argparse
import math
import sys
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import gen_data as gd
from .cnn_op import NATIVE_ARGS, OpTune, add_codegen_annotations
from .digest import Digest, KNOWN_SEEDS, OpRun, OpTuneWisdom, OpWisdom, SsdsDiff, read_wisdoms, write_wisdoms
from .op import Dims, Op, RtErr, UnsupErr, parse_lexp, read_ops
from .rtc import HipCompute, RtcArg, RtcCompileOpts, RtcFuncCall, RtcFuncInfo

MRD_TOLER = 2e-4  # src/rtc_prof.cc:161
PEAK_FP32_MFMA = 157.3e12
PEAK_HBM = 8.0e12


@dataclass
class PrcRet:
    op: Op          # the annotated op that ran
    rt_secs: float  # duration of the LAST of run_iter launches (reference behaviour, src/rtc_prof.cc:107,123)
    all_secs: List[float]
    launch: dict


class OpsBackend:
    """One initialised backend + its compiled generator functions (rtc + codegen pair of src/rtc_prof.cc:130-136)."""

    def __init__(self, rtc: HipCompute, compile_opts: Optional[RtcCompileOpts] = None):
        self.rtc = rtc
        self.compile_opts = compile_opts or RtcCompileOpts()
        if not getattr(rtc, "_gen_data_compiled", False):  # one generator module per backend instance
            self.rtc.compile(gd.func_infos(), self.compile_opts)
            rtc._gen_data_compiled = True
        self._fn_ix = 0
        self._gen_fns: Dict[str, str] = {}  # annotated-op text -> generated function name (rtc_func_sigs_map analogue)

    def gen_func(self, anno_op: Op) -> str:
        """Unique function per distinct annotated op (src/rtc_func_gen.cc:590-621); native funcs carry a stub body."""
        key = anno_op.to_str()
        if key in self._gen_fns:
            return self._gen_fns[key]
        fn = anno_op.get_func_name()
        if fn not in NATIVE_ARGS:
            raise UnsupErr(f"no template for function {fn!r} in this backend")
        gen_fn = f"{fn}__{self._fn_ix}"
        self._fn_ix += 1
        args = [a for a, _ in NATIVE_ARGS[fn]]
        stub = f"CUCL_GLOBAL_KERNEL void {gen_fn}( void ) {{ }}\n"  # never launched: the backend binds by op.func_name
        self.rtc.compile([RtcFuncInfo(gen_fn, stub, args, anno_op)], self.compile_opts)
        self._gen_fns[key] = gen_fn
        return gen_fn

    def gc_clear(self):
        for gen_fn in self._gen_fns.values():
            self.rtc.release_func(gen_fn)
        self._gen_fns.clear()


def profile_rcg_call(be: OpsBackend, anno_op: Op, gen_mode: Optional[int], gen_vi: float = 0.0, run_iter: int = 1,
                     want_outs: bool = True, include_ins: bool = False, tile: str = "") -> Tuple[Dict[str, np.ndarray], PrcRet]:
    rtc = be.rtc
    fn = anno_op.get_func_name()
    gen_fn = be.gen_func(anno_op)
    arg_map: Dict[str, RtcArg] = {}
    created: List[str] = []
    outs: Dict[str, np.ndarray] = {}
    tune_key = "sgemm_tile" if anno_op.get_type() == "sgemm" else "conv_tile"
    # an arg whose annotated dims differ from its reference dims (<arg>_ref) lives twice: data is generated in / read back from the
    # reference-layout var "<arg>_ref", a layout pass (run outside the timed call) fills / reads the kernel's var (src/rtc_prof.cc:92-121)
    def ref_dims_of(an):
        return anno_op.get_dims(an + "_ref") if anno_op.has(an + "_ref") else anno_op.get_dims(an)
    try:
        for an, io in NATIVE_ARGS[fn]:
            if io == "REF":
                arg_map[an] = RtcArg.ref(anno_op.get_dims(an))
                continue
            dims = anno_op.get_dims(an)
            rtc.create_var_with_dims(an, dims)  # zero-filled
            created.append(an)
            arg_map[an] = RtcArg.var(an)
            if ref_dims_of(an) != dims:
                rtc.create_var_with_dims(an + "_ref", ref_dims_of(an)); created.append(an + "_ref")
        if any(anno_op.has(an + "_ref") for an, _ in NATIVE_ARGS[fn]):
            from . import nhwc
            nhwc.ensure_compiled(rtc)
        if gen_mode is not None:
            for an, io in NATIVE_ARGS[fn]:
                if io != "IN":
                    continue
                dims, rdims = anno_op.get_dims(an), ref_dims_of(an)
                gen_vn = an if rdims == dims else an + "_ref"
                rtc.run(gd.gen_call(anno_op.get_type(), an, gen_vn, rdims, gen_mode, gen_vi))
                if gen_vn != an:
                    rtc.run(nhwc.xpose_call(an, gen_vn, an, rdims, dims, anno_op))
                if include_ins and want_outs:
                    outs[an] = rtc.create_nda_from_var(gen_vn)
        rtc.set_tune(tune_key, tile)
        rfc = RtcFuncCall(gen_fn, arg_map)
        ids = [rtc.run(rfc) for _ in range(run_iter)]
        launch = rtc.last_launch()
        if want_outs:
            for an, io in NATIVE_ARGS[fn]:
                if io == "OUT":
                    dims, rdims = anno_op.get_dims(an), ref_dims_of(an)
                    if rdims != dims:
                        rtc.run(nhwc.xpose_call(an, an + "_ref", an, rdims, dims, anno_op))
                        outs[an] = rtc.create_nda_from_var(an + "_ref")
                    else:
                        outs[an] = rtc.create_nda_from_var(an)
        rtc.finish_and_sync()
        secs = [rtc.get_dur(i, i) / 1000.0 for i in ids]
        return outs, PrcRet(anno_op, secs[-1], secs, launch)
    finally:
        rtc.set_tune(tune_key, "")
        rtc.finish_and_sync()
        for vn in created:
            rtc.release_var(vn)
        rtc.release_per_call_id_data()
        be.gc_clear()


def eff_row(op: Op, secs: float) -> dict:
    fl, by = op.flops(), op.algo_bytes()
    ai = fl / by
    bound = "mfma" if ai > PEAK_FP32_MFMA / PEAK_HBM else "hbm"
    return {"flops": fl, "bytes": by, "ai": ai, "secs": secs, "tflops": fl / secs / 1e12, "gbs": by / secs / 1e9,
            "bound": bound, "frac_mfma": fl / secs / PEAK_FP32_MFMA, "frac_hbm": by / secs / PEAK_HBM}


def ops_prof(rtc: HipCompute, ops: List[Op], op_tunes: Dict[str, OpTune], kg_tune_tag: str, gen_mode: Optional[int] = 5,
             gen_vi: float = 0.0, run_iter: int = 1, wisdom_in: Optional[List[OpWisdom]] = None, mrd_toler: float = MRD_TOLER,
             func_mrd_toler: Optional[Dict[str, float]] = None, write_kg_digest: bool = True, write_runs: bool = False,
             out=sys.stdout, max_err: int = 10) -> Tuple[List[OpWisdom], int, List[dict]]:
    """-> (output wisdoms, num_mad_fail, per-(op,tune) result rows)."""
    if kg_tune_tag not in op_tunes:
        raise RtErr(f"kg_tune_tag {kg_tune_tag!r} is not one of the op_tunes")
    be = OpsBackend(rtc)
    plat_tag = rtc.get_plat_tag()
    order = [kg_tune_tag] + [t for t in op_tunes if t != kg_tune_tag]
    num_mad_fail = 0
    wout: List[OpWisdom] = []
    rows: List[dict] = []
    for op_ix, op in enumerate(ops):
        win = wisdom_in[op_ix] if wisdom_in is not None else None
        if win is not None and win.op != op:
            raise RtErr(f"op mismatch between input wisdom and ops-list: {win.op.to_str()} vs {op.to_str()}")
        ow = OpWisdom(op)
        seen_err = False
        vs_kg: Optional[Dict[str, np.ndarray]] = None
        for tag in order:
            tune = op_tunes[tag]
            otw = OpTuneWisdom(tune.to_str())
            err, err_extra, prc, vsi = "", "", None, None
            try:
                anno = add_codegen_annotations(op, tune)
            except UnsupErr as e:
                err = "annotation failure: " + str(e)
            if not err:
                try:
                    vsi, prc = profile_rcg_call(be, anno, gen_mode, gen_vi, run_iter, tile=tune.hip_tile)
                except UnsupErr as e:
                    err = "profile call failure: " + str(e)
            if tag == kg_tune_tag:
                if err:
                    err += f"known-good op_tune (kg_tune_tag={kg_tune_tag}) failed. Can't write digests or do live comparisons."
                else:
                    vs_kg = vsi
                    if write_kg_digest:
                        for vn in sorted(vs_kg):
                            seed = KNOWN_SEEDS.get(vn, 0x9E3779B97F4A7C15)
                            ow.kgs.append((vn, Digest.from_array(vs_kg[vn], op.get_dims(vn), seed)))
            if not err:
                fnm = prc.op.get_func_name()
                vmt = (func_mrd_toler or {}).get(fnm, mrd_toler)
                if vs_kg is not None:
                    if sorted(vs_kg) != sorted(vsi):
                        raise RtErr("reg/comp out var set mismatch")
                    for vn in sorted(vs_kg):
                        sd = SsdsDiff.of(vs_kg[vn], vsi[vn])
                        if sd.has_nan() or sd.mrd >= vmt:
                            num_mad_fail += 1
                            err += f"{vn}: {op.get_dims(vn).pretty()} ssds_str(out_batch_1,out_batch_2)={sd.basic_str()}"
                if win is not None:
                    if len(win.kgs) != len(vsi):
                        raise RtErr("digest count mismatch vs input wisdom")
                    for (vn, kg), vn2 in zip(win.kgs, sorted(vsi)):
                        if vn != vn2:
                            raise RtErr("digest var name mismatch vs input wisdom")
                        dg = Digest.from_array(vsi[vn], op.get_dims(vn), kg.seed)
                        comp = kg.mrd_comp(dg, vmt)
                        if comp:
                            err += f"{vn} digest mrd_comp() failure vs stored digest"
                            err_extra += "comp_res:\n" + comp + "\n"
                else:
                    err_extra += ""  # reference records 'skipped, no input wisdom' as an error string; we only note it
            otw.runs[plat_tag] = OpRun(plat_tag, prc.rt_secs if prc else float("nan"), err, prc.op if (prc and not err) else None)
            ow.wisdoms.append(otw)
            row = {"op_ix": op_ix, "tune": tag, "err": err}
            if prc is not None:
                row.update(eff_row(op, prc.rt_secs)); row["launch"] = prc.launch; row["all_secs"] = prc.all_secs
            rows.append(row)
            if err:
                if not seen_err:
                    print(f"-----\n errors for op_ix={op_ix} op='{op.to_str()}'", file=out)
                    seen_err = True
                print(f"--  comp fail for op_tune='{tune.to_str()}'\n{err}\n{err_extra}", end="", file=out)
        if not write_runs:
            ow.wisdoms = []
        if not write_kg_digest and win is not None:
            ow.kgs = win.kgs
        wout.append(ow)
    rtc.finish_and_sync()
    print("***ALL IS WELL***" if not num_mad_fail else f"***MAD FAILS*** num_mad_fail={num_mad_fail}", file=out)
    return wout, num_mad_fail, rows


def _parse_tunes(s: str) -> Dict[str, OpTune]:
    items = parse_lexp(s)
    if isinstance(items, str):
        raise RtErr("--op-tunes must be a list: '(tag=(k=v,...),...)'")
    out = {}
    for tag, v in items:
        inner = "(" + ",".join(f"{k}={vv}" for k, vv in (v if not isinstance(v, str) else [])) + ")"
        out[tag] = OpTune.parse(inner)
    return out


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="boda_amd.ops_prof", description=__doc__.split("\n")[0])
    ap.add_argument("--ops-fn", required=True)
    ap.add_argument("--op-tunes", default="(def=())")
    ap.add_argument("--kg-tune-tag", default="def")
    ap.add_argument("--gen-data-mode", type=int, default=5)
    ap.add_argument("--gen-data-vi", type=float, default=0.0)
    ap.add_argument("--run-iter", type=int, default=1)
    ap.add_argument("--wisdom-in-fn")
    ap.add_argument("--wisdom-out-fn")
    ap.add_argument("--write-runs", type=int, default=0)
    ap.add_argument("--mrd-toler", type=float, default=MRD_TOLER)
    ap.add_argument("--rtc", default="(be=hip)")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--eff", type=int, default=1, help="print per-op efficiency rows (cnn_op_info style)")
    a = ap.parse_args(argv)
    from .rtc import make_rtc
    rtc = make_rtc(a.rtc, a.device)
    rtc.init()
    ops = read_ops(a.ops_fn)
    win = read_wisdoms(a.wisdom_in_fn) if a.wisdom_in_fn else None
    wout, nfail, rows = ops_prof(rtc, ops, _parse_tunes(a.op_tunes), a.kg_tune_tag, a.gen_data_mode, a.gen_data_vi, a.run_iter, win,
                                 a.mrd_toler, write_runs=bool(a.write_runs))
    if a.eff:
        for r in rows:
            if "tflops" in r:
                print(f"op {r['op_ix']:3d} {r['tune']:>10s}  {r['secs']*1e3:9.4f} ms  {r['tflops']:8.2f} TF/s ({100*r['frac_mfma']:5.1f}% mfma)  "
                      f"{r['gbs']:8.1f} GB/s ({100*r['frac_hbm']:5.1f}% hbm)  AI {r['ai']:7.1f}  {r['launch']['cfg']} grid {r['launch']['grid']}")
    if a.wisdom_out_fn:
        write_wisdoms(a.wisdom_out_fn, wout)
    return 1 if nfail else 0


if __name__ == "__main__":
    sys.exit(main())
