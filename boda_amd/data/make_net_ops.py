#!/usr/bin/env python3
"""Generates boda_amd/data/ops/{googlenet_conv,resnet-50}-conv-ops-b1.txt and boda_amd/data/nets/* from the reference's net definitions
(nets/<net>/train_val.prototxt) with THIS project's prototxt reader + shape inference (boda_amd/prototxt.py).  Run in the
build container only.  The outputs are data: one op line per Convolution layer, batch 1 (re-batched by the callers);
layer order and multiplicity preserved (64 GoogLeNet convs incl. the auxiliary heads, 53 ResNet-50 convs + its fc)."""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(os.path.dirname(HERE))  # boda_amd/data -> repo root; sys.path.insert(0, ROOT)
from boda_amd.prototxt import conv_bottoms, conv_ops, pipe_spec
REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
for net in ("googlenet_conv", "resnet-50"):
    ops = conv_ops(open(os.path.join(REF, "nets", net, "train_val.prototxt")).read(), 1)
    with open(os.path.join(HERE, "ops", f"{net}-conv-ops-b1.txt"), "w") as f:
        for name, op in ops:
            f.write(op.to_str() + "\n")
    print(net, len(ops), "convs", sum(o.flops() for _, o in ops) / 1e9, "GF at batch 1")
    # which blob every conv of that list reads (name + bottom, same order): sibling convolutions = same bottom (bench.py --group-siblings)
    os.makedirs(os.path.join(HERE, "nets"), exist_ok=True)
    cb = conv_bottoms(open(os.path.join(REF, "nets", net, "train_val.prototxt")).read())
    assert [n for n, _ in cb] == [n for n, _ in ops]
    with open(os.path.join(HERE, "nets", f"{net}-conv-bottoms.txt"), "w") as f:
        f.write("".join(f"{n} {b}\n" for n, b in cb))

# full-net op records (TEST phase) for the nets whose every layer type has a forward kernel here (GoogLeNet: conv, ReLU, max/avg
# pool, LRN, Concat, Dropout) -- read back by boda_amd.conv_pipe.pipe_from_spec
os.makedirs(os.path.join(HERE, "nets"), exist_ok=True)
for net in ("googlenet_conv",):
    spec = pipe_spec(open(os.path.join(REF, "nets", net, "train_val.prototxt")).read())
    with open(os.path.join(HERE, "nets", f"{net}.txt"), "w") as f:
        f.write("# TEST-phase forward ops of nets/%s/train_val.prototxt, one record per op (boda_amd/prototxt.py:pipe_spec)\n" % net)
        f.write("\n".join(spec) + "\n")
    print(net, len(spec), "records")
