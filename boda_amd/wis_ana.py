"""`wis-ana` mode: analyse a wisdom file -- per-op best tune, best single tune over all ops, a reference tune -- and write the CSV the
reference's plotting script reads (src/op-tuner.cc:204-392; behaviour restated, not code).

    python -m boda_amd.wis_ana --wisdom-in-fn w.wis [--csv-out-fn out.csv] [--ops-out-fn ops.tex [--ops-out-brief 1]] [--s-img N] [--s-plat REGEX]
                               [--ref-tune '(use_be=nvrtc,use_culibs=1)'] [--min-flops F] [--show-aom 0] [--show-pom 0] [--show-ref 0]
                               [--aom-tag T] [--pom-tag T] [--ref-tag T] [--verbose 1] [--tile-wisdom-out-fn tiles.txt]

CSV: header `OP FLOPS [<aom-tag>] [<pom-tag>] [<ref-tag>]`, then one row per op in op order (op_base_t::operator<): the op line, its
2*M*N*K, and seconds -- AOM = the one tune that ran the most ops (ties: least total time) ("all-ops manual"), POM = the per-op minimum over
every tune but the reference tune ("per-op / autotuned"), REF = the reference tune's run; `nan` where there is no such run.  Runs with an
error, or on a platform --s-plat does not match, are dropped first.  Prints `tot_runs=<n>` (runs that entered the AOM / POM selection).

The same per-op minimum closes the tuning loop on this backend: --tile-wisdom-out-fn writes, for every op whose fastest run carried an
op_tune with a `hip_tile`, the line `<op> TAB <tile> TAB <secs>`; boda_amd.cnn_op.add_codegen_annotations(op, tune, tile_wisdom=...) then
gives that tile to the op's function (it travels with the function as str_val hip_tile and overrides the native planner's cost model for
that function only -- the op_tune_t-per-op analogue of the reference's wisdom-driven tuning)."""
from __future__ import annotations
import argparse
import math
import re
import sys
from typing import Dict, List, Optional, TextIO, Tuple

from .digest import OpRun, OpWisdom, read_wisdoms
from .op import Op, RtErr, parse_lexp, parse_op
from .wis_merge import op_ref_key


def fmt_g(v: float) -> str:
    """What `std::ostream << double` prints at the default precision (%g with 6 significant digits; nan as `nan`)."""
    if math.isnan(v):
        return "nan"
    return "%g" % v


def get_op_flops(op: Op) -> int:
    """wis_ana_t::get_op_flops (src/op-tuner.cc:238-262): Convolution only -- M = all images' output positions, K = in_chan * y * x, N = out_chan."""
    dout, din, f = op.get_dims("out"), op.get_dims("in"), op.get_dims("filts")
    if din.dsz("img") != dout.dsz("img"):
        raise RtErr("wis-ana: in and out disagree in img")
    return dout.dsz("img") * dout.dsz("x") * dout.dsz("y") * f.dsz("in_chan") * f.dsz("x") * f.dsz("y") * f.dsz("out_chan") * 2


class PerOpAna:
    def __init__(self):
        self.min_r: Optional[OpRun] = None; self.min_tune: str = ""; self.ref_r: Optional[OpRun] = None


def wis_ana(ows: List[OpWisdom], s_img: int = 0, s_plat: str = ".*", ref_tune: Optional[str] = None, min_flops: float = 0.0,
            show_aom: bool = True, show_pom: bool = True, show_ref: bool = True, aom_tag: str = "boda-manual-tune", pom_tag: str = "boda-autotuned",
            ref_tag: str = "REF", csv_out: Optional[TextIO] = None, ops_out: Optional[TextIO] = None, ops_out_brief: bool = False,
            verbose: bool = False, out: TextIO = sys.stdout) -> Tuple[int, List[Tuple[OpWisdom, PerOpAna]], str]:
    """-> (tot_runs, [(op wisdom, its analysis)] in op order, the all-ops-best tune text)."""
    if csv_out is not None:
        csv_out.write("OP FLOPS" + (" " + aom_tag if show_aom else "") + (" " + pom_tag if show_pom else "") + (" " + ref_tag if show_ref else "") + "\n")
    r_plat = re.compile(s_plat)
    all_wis: Dict[tuple, OpWisdom] = {}
    for ow in ows:
        if s_img and ow.op.get_dims("in").dsz("img") != s_img:
            continue
        if not (get_op_flops(ow.op) >= min_flops):
            continue
        k = op_ref_key(ow.op)
        if k in all_wis:
            raise RtErr("wis-ana: the same op twice in one wisdom file (merge first: wis-merge)")
        # filter_runs: no errors, platform must match (regex_search)
        kept = OpWisdom(ow.op, [], [type(t)(t.op_tune, {tag: r for tag, r in t.runs.items() if not r.err and r_plat.search(r.be_plat_tag)}) for t in ow.wisdoms])
        all_wis[k] = kept
    order = [all_wis[k] for k in sorted(all_wis)]
    tot_runs = 0
    filt_scores: Dict[str, List[float]] = {}      # tune text -> [total seconds, number of runs]
    anas: List[PerOpAna] = []
    for ow in order:
        poa = PerOpAna(); min_time = float("inf")
        for t in ow.wisdoms:
            for tag in sorted(t.runs):
                r = t.runs[tag]
                if ref_tune is not None and t.op_tune == ref_tune:
                    if poa.ref_r is not None:
                        raise RtErr("wis-ana: more than one run of the reference tune for an op (filter to one platform with --s-plat)")
                    poa.ref_r = r
                else:
                    tot_runs += 1
                    fs = filt_scores.setdefault(t.op_tune, [0.0, 0]); fs[0] += r.rt_secs; fs[1] += 1
                    if r.rt_secs < min_time:
                        min_time = r.rt_secs; poa.min_r = r; poa.min_tune = t.op_tune
        anas.append(poa)
    # the one tune for all ops: most runs first, least total time second (first in tune-text order on a full tie)
    min_filt_tune, best = "", None
    for tune in sorted(filt_scores):
        secs, num = filt_scores[tune]
        if best is None or num > best[1] or (num == best[1] and secs < best[0]):
            best = (secs, num); min_filt_tune = tune
    for ow, poa in zip(order, anas):
        if verbose:
            out.write(f"owi->op={ow.op.to_str()}\n")
        if csv_out is not None:
            csv_out.write(f"{ow.op.to_str()} {get_op_flops(ow.op)}")
        if ops_out is not None:
            from .cnn_op_info import OpInfoToLatex
            ops_out.write(OpInfoToLatex(ow.op, 2, 1, False).info_row(ops_out_brief))
        if show_aom:
            v = float("nan")
            for t in ow.wisdoms:
                for tag in sorted(t.runs):
                    if t.op_tune == min_filt_tune:
                        v = t.runs[tag].rt_secs
                        if verbose:
                            out.write(f"  ALL-OP MIN: r.be_plat_tag={tag} r.rt_secs={fmt_g(v)} min_tune={t.op_tune}\n")
            if csv_out is not None:
                csv_out.write(" " + fmt_g(v))
        if show_pom:
            v = poa.min_r.rt_secs if poa.min_r is not None else float("nan")
            if verbose and poa.min_r is not None:
                out.write(f"  PER-OP MIN: r.be_plat_tag={poa.min_r.be_plat_tag} r.rt_secs={fmt_g(v)} min_tune={poa.min_tune}\n")
            if csv_out is not None:
                csv_out.write(" " + fmt_g(v))
        if show_ref:
            v = poa.ref_r.rt_secs if poa.ref_r is not None else float("nan")
            if verbose and poa.ref_r is not None:
                out.write(f"  PER-OP REF: r.be_plat_tag={poa.ref_r.be_plat_tag} r.rt_secs={fmt_g(v)} min_tune={ref_tune}\n")
            if csv_out is not None:
                csv_out.write(" " + fmt_g(v))
        if csv_out is not None:
            csv_out.write("\n")
    out.write(f"tot_runs={tot_runs}\n")
    return tot_runs, list(zip(order, anas)), min_filt_tune


# ---- the per-op best-tile table (what closes the tuning loop on this backend) -----------------------------------------------------
def tile_of_tune(op_tune: str) -> str:
    items = parse_lexp(op_tune) if op_tune.strip() else []
    return next((str(v) for k, v in (items if not isinstance(items, str) else []) if k == "hip_tile"), "")


class TileWisdom:
    """op line (canonical text) -> the tile of the op's fastest recorded run.  Text form: `<op> TAB <tile> TAB <secs>` per line."""

    def __init__(self, table: Optional[Dict[str, Tuple[str, float]]] = None):
        self.table: Dict[str, Tuple[str, float]] = dict(table or {})

    @staticmethod
    def from_analysis(rows: List[Tuple[OpWisdom, PerOpAna]]) -> "TileWisdom":
        tw = TileWisdom()
        for ow, poa in rows:
            if poa.min_r is not None and tile_of_tune(poa.min_tune):
                tw.table[ow.op.to_str()] = (tile_of_tune(poa.min_tune), poa.min_r.rt_secs)
        return tw

    @staticmethod
    def from_wisdoms(ows: List[OpWisdom], s_plat: str = ".*") -> "TileWisdom":
        import io
        _, rows, _ = wis_ana(ows, s_plat=s_plat, out=io.StringIO())
        return TileWisdom.from_analysis(rows)

    @staticmethod
    def load(path: str) -> "TileWisdom":
        tw = TileWisdom()
        with open(path) as f:
            for line in f:
                if line.strip():
                    parts = line.rstrip("\n").split("\t")
                    if len(parts) < 2:
                        raise RtErr(f"tile wisdom: bad line {line[:60]!r}")
                    tw.table[parse_op(parts[0]).to_str()] = (parts[1], float(parts[2]) if len(parts) > 2 else float("nan"))
        return tw

    def save(self, path: str) -> None:
        with open(path, "w") as f:
            for k in sorted(self.table):
                f.write(f"{k}\t{self.table[k][0]}\t{self.table[k][1]!r}\n")

    def tile_for(self, op: Op) -> str:
        """The recorded tile for this op ('' if none); `op` is the UN-annotated op (the key wisdom files use)."""
        return self.table.get(op.to_str(), ("", 0.0))[0]


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="boda_amd.wis_ana", description="analyse a wisdom file (the reference's wis-ana mode)")
    ap.add_argument("--wisdom-in-fn", required=True); ap.add_argument("--csv-out-fn"); ap.add_argument("--ops-out-fn"); ap.add_argument("--ops-out-brief", type=int, default=0)
    ap.add_argument("--verbose", type=int, default=0); ap.add_argument("--s-img", type=int, default=0); ap.add_argument("--s-plat", default=".*")
    ap.add_argument("--ref-tune"); ap.add_argument("--min-flops", type=float, default=0.0)
    ap.add_argument("--show-aom", type=int, default=1); ap.add_argument("--aom-tag", default="boda-manual-tune")
    ap.add_argument("--show-pom", type=int, default=1); ap.add_argument("--pom-tag", default="boda-autotuned")
    ap.add_argument("--show-ref", type=int, default=1); ap.add_argument("--ref-tag", default="REF")
    ap.add_argument("--tile-wisdom-out-fn", help="also write the per-op best-tile table (ops whose fastest run's op_tune carries a hip_tile)")
    a = ap.parse_args(argv)
    files = [open(fn, "w") if fn else None for fn in (a.csv_out_fn, a.ops_out_fn)]
    try:
        _, rows, _ = wis_ana(read_wisdoms(a.wisdom_in_fn), a.s_img, a.s_plat, a.ref_tune, a.min_flops, bool(a.show_aom), bool(a.show_pom), bool(a.show_ref),
                             a.aom_tag, a.pom_tag, a.ref_tag, files[0], files[1], bool(a.ops_out_brief), bool(a.verbose))
    finally:
        for f in files:
            if f:
                f.close()
    if a.tile_wisdom_out_fn:
        TileWisdom.from_analysis(rows).save(a.tile_wisdom_out_fn)
    return 0


if __name__ == "__main__":
    sys.exit(main())
