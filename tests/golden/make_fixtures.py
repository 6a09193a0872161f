#!/usr/bin/env python3
"""Regenerates tests/golden/ from the reference checkout (run in the build container only).

Fixtures are DATA held by the reference's own tests -- known-answer digests (wisdom files) and the
op lists they were produced from -- copied byte-for-byte.  No reference source code is copied.
`/root/reference` does not exist on the GPU box; tests read only the committed copies.

  wisdom/<name>.wis   <- test/good_tr/<name>/wisdom.wis      (expected outputs: nda_digest_t hex)
  wisdom/wisdom-merged-head3.wis <- first 3 records of test/wisdom-merged.wis (run-record format)
  ops/<file>          <- test/<file>                          (inputs: one op per line)
"""
import os, shutil, sys
REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
WIS = ["sgemm-gen5", "sgemm-gen600", "conv-gen5", "conv-debug", "conv-full-gen5", "ops-prof-conv-3x3-cudnn-boda"]
OPS = ["sgemm-ops-micro.txt", "sgemm-ops-tiny.txt", "sgemm-ops-small.txt", "sgemm-ops-full.txt", "sgemm-ops-debug.txt", "sgemm-ops-debug-half.txt",
       "conv-ops-debug.txt", "conv-ops-debug-tmp.txt", "conv-ops-tiny.txt", "conv-ops-small.txt",
       "conv-ops-1-5-20-nin-alex-gn.txt", "ops/conv/conv-ops-kern-3x3-batch-1-5-20-nin-alex-gn.txt"]
os.makedirs(os.path.join(HERE, "wisdom"), exist_ok=True)
os.makedirs(os.path.join(HERE, "ops"), exist_ok=True)
for w in WIS:
    shutil.copyfile(os.path.join(REF, "test/good_tr", w, "wisdom.wis"), os.path.join(HERE, "wisdom", w + ".wis"))
for o in OPS:
    shutil.copyfile(os.path.join(REF, "test", o), os.path.join(HERE, "ops", os.path.basename(o)))
# excerpt (first 3 op records) of test/wisdom-merged.wis: the only fixture holding op_tune_wisdom_t / op_run_t records
n, out = 0, []
for l in open(os.path.join(REF, "test/wisdom-merged.wis")):
    out.append(l)
    if l.strip() == "/op_wisdom_t":
        n += 1
        if n == 3: break
open(os.path.join(HERE, "wisdom", "wisdom-merged-head3.wis"), "w").writelines(out)
print("ok")
