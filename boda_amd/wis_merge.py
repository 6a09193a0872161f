"""`wis-merge` mode: merge several wisdom files into one (src/op-tuner.cc:126-182; behaviour restated, not code).

    python -m boda_amd.wis_merge --wisdom-out-fn merged.wis [--keep-kgs 1] a.wis b.wis ...

* ops are keyed by op_base_t::operator< (src/op_base.cc:16-23: the str_vals map, then the nda_vals map, each nda by its dims -- type
  name first, then the (size, stride, name) triples -- then null-ness, then elements); the output is SORTED by op, not in input order;
* the op_tune_wisdom_t records of one op are keyed by the op_tune's printed text (src/op-tuner.H:30-35) and come out sorted by it;
* runs of the same (op, op_tune) from different files are united; the same platform tag twice is an error (`must_insert`,
  src/op-tuner.cc:127-132 -- "for now, don't allow overwrite");
* known-good digests (`kg` records) cannot be merged: dropped unless --keep-kgs 1, which keeps those of the FIRST file that has the op.
"""
from __future__ import annotations
import argparse
import sys
from typing import Dict, List

from .digest import OpTuneWisdom, OpWisdom, read_wisdoms, write_wisdoms
from .op import Nda, Op, RtErr


def _nda_ref_key(n: Nda):
    """nda_t::operator< (src/boda_base.cc:455-463) over dims_t::operator< (src/boda_base.H:521-522, dim_t :435-440)."""
    d = n.dims
    dims = tuple(zip(d.sizes, d.strides, d.names)) if d is not None else ()
    return (n.tn, dims, n.v is not None, tuple(n.v) if n.v is not None else ())


def op_ref_key(op: Op):
    """op_base_t::operator< as a sort key (src/op_base.cc:16-23)."""
    return (tuple(sorted(op.str_vals.items())), tuple((k, _nda_ref_key(op.nda_vals[k])) for k in sorted(op.nda_vals)))


def merge_tune_wisdoms(a: List[OpTuneWisdom], b: List[OpTuneWisdom]) -> List[OpTuneWisdom]:
    """op_wisdom_t::merge_wisdoms_from: one record per op_tune text, sorted by it; runs united, a duplicate platform tag is an error."""
    by_tune: Dict[str, OpTuneWisdom] = {}
    for otw in list(a) + list(b):
        have = by_tune.get(otw.op_tune)
        if have is None:
            by_tune[otw.op_tune] = OpTuneWisdom(otw.op_tune, dict(otw.runs))
            continue
        for tag, run in otw.runs.items():
            if tag in have.runs:
                raise RtErr(f"wis-merge: op_tune {otw.op_tune} already has a run on platform {tag!r} (runs are never overwritten)")
            have.runs[tag] = run
    return [by_tune[k] for k in sorted(by_tune)]


def merge_wisdoms(files: List[List[OpWisdom]], keep_kgs: bool = False) -> List[OpWisdom]:
    all_wis: Dict[tuple, OpWisdom] = {}
    for ows in files:
        for ow in ows:
            k = op_ref_key(ow.op)
            have = all_wis.get(k)
            if have is None:
                all_wis[k] = OpWisdom(ow.op, list(ow.kgs) if keep_kgs else [], list(ow.wisdoms))
            else:
                have.wisdoms = merge_tune_wisdoms(have.wisdoms, ow.wisdoms)
    return [all_wis[k] for k in sorted(all_wis)]


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="boda_amd.wis_merge", description="merge wisdom files (the reference's wis-merge mode)")
    ap.add_argument("wisdom_in_fns", nargs="+"); ap.add_argument("--wisdom-out-fn", required=True); ap.add_argument("--keep-kgs", type=int, default=0)
    a = ap.parse_args(argv)
    out = open(a.wisdom_out_fn, "w"); out.close()       # (opened early, as the reference does, to fail before the work)
    merged = merge_wisdoms([read_wisdoms(fn) for fn in a.wisdom_in_fns], bool(a.keep_kgs))
    write_wisdoms(a.wisdom_out_fn, merged)
    print(f"merged {len(a.wisdom_in_fns)} files: {len(merged)} ops, {sum(len(t.runs) for ow in merged for t in ow.wisdoms)} runs")
    return 0


if __name__ == "__main__":
    sys.exit(main())
