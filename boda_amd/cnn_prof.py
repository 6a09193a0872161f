"""`cnn_prof` mode: op list -> annotated function signatures, no device (BASELINE configs[0], "plumbing, no GPU").

Restates cnn_prof_t::main (src/cnn-prof.cc:144-156): each input line is parsed, annotated with the given op_tune
(variant selection: which function runs the op, here the native side door of be=hip), given conv_has_relu=1 (Convolution)
and written back as one annotated op line.  sgemm lines go through add_codegen_annotations' sgemm branch
(src/cnn_op.cc:338-378).  Runs on the CPU only.
    python -m boda_amd.cnn_prof --cnn-func-sigs-fn tests/golden/ops/sgemm-ops-tiny.txt [--op-tune '(use_culibs=1)'] [--rtc-func-sigs-fn out.txt]
"""
from __future__ import annotations
import argparse
import sys
from typing import List

from .cnn_op import OpTune, add_codegen_annotations
from .op import Op, UnsupErr, read_ops


def cnn_prof(ops: List[Op], op_tune: OpTune) -> List[str]:
    out = []
    for op in ops:
        try:
            out.append(add_codegen_annotations(op, op_tune).to_str())
        except UnsupErr as e:
            out.append("# unsupported: " + str(e))
    return out


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="boda_amd.cnn_prof", description=__doc__.split("\n")[0])
    ap.add_argument("--cnn-func-sigs-fn", required=True, help="file to read ops from (one op line per line, current or legacy form)")
    ap.add_argument("--rtc-func-sigs-fn", default="-", help="output: annotated op per line ('-' = stdout)")
    ap.add_argument("--op-tune", default="()")
    a = ap.parse_args(argv)
    lines = cnn_prof(read_ops(a.cnn_func_sigs_fn), OpTune.parse(a.op_tune))
    f = sys.stdout if a.rtc_func_sigs_fn == "-" else open(a.rtc_func_sigs_fn, "w")
    for l in lines:
        f.write(l + "\n")
    return 0


if __name__ == "__main__":
    sys.exit(main())
