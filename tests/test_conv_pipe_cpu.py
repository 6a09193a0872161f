"""conv_pipe shape inference (CPU): NiN / AlexNet node dims and conv flop totals match the reference's accounting."""
from boda_amd.conv_pipe import nin_imagenet, alexnet_ng_conv
import bench


def test_nin_shapes_and_flops():
    cp = nin_imagenet(256)
    assert cp.nodes["conv1"].sizes == (256, 96, 55, 55) and cp.nodes["pool0"].sizes == (256, 96, 27, 27)
    assert cp.nodes["pool2"].sizes == (256, 256, 13, 13) and cp.nodes["pool3"].sizes == (256, 384, 6, 6)
    assert cp.nodes["cccp8"].sizes == (256, 1000, 6, 6) and cp.nodes["pool4"].sizes == (256, 1000, 1, 1)
    assert cp.conv_flops() == sum(o.flops() for o in bench.nin_ops(256))  # 566.3 GF
    assert abs(cp.conv_flops() / 1e9 - 566.3) < 0.1


def test_alexnet_shapes_and_flops():
    cp = alexnet_ng_conv(256)
    assert cp.nodes["pool1"].sizes == (256, 96, 27, 27) and cp.nodes["pool5"].sizes == (256, 256, 6, 6)
    assert cp.nodes["fc6"].sizes == (256, 4096, 1, 1) and cp.nodes["fc8"].sizes == (256, 1000, 1, 1)
    assert cp.conv_flops() == sum(o.flops() for o in bench.alexnet_b256_ops(256))  # 581.3 GF


def test_googlenet_from_spec_fixture():
    """pipe_from_spec over the GoogLeNet op records: 64 convs whose op lines equal the per-layer conv-ops fixture (same reader,
    two routes), inception Concat channel sums, ceil-mode pools, InnerProduct-free classifier heads."""
    import os
    from boda_amd.conv_pipe import googlenet_conv
    from boda_amd.op import parse_op
    cp = googlenet_conv(1)
    convs = [o for o in cp.ops if o.type == "Convolution"]
    assert len(convs) == 64 and sum(o.type == "Concat" for o in cp.ops) == 9 and sum(o.type == "LRN" for o in cp.ops) == 2
    here = os.path.dirname(os.path.abspath(__file__))
    want = [parse_op(l) for l in open(os.path.join(os.path.dirname(here), "boda_amd", "data", "ops", "googlenet_conv-conv-ops-b1.txt")).read().splitlines() if l.strip()]
    assert [cp.conv_op(o).to_str() for o in convs] == [w.to_str() for w in want]
    assert cp.nodes["pool1"].sizes == (1, 64, 56, 56) and cp.nodes["icp2_out"].sizes == (1, 480, 28, 28) and cp.nodes["icp9_out"].sizes == (1, 1024, 7, 7)
    assert cp.nodes[cp.out_node()].sizes == (1, 1000, 1, 1)
    assert abs(cp.conv_flops() / 1e9 - 3.182) < 1e-3


def test_sibling_runs_split_where_the_shared_bottom_is_rewritten():
    """Sibling fusion emits a group at its first member's place; a member behind an in-place op on the shared bottom must not be hoisted above it."""
    from boda_amd.conv_pipe import ConvPipe, PipeOp, sibling_runs
    from boda_amd.op import Dims
    p = ConvPipe("t", "data", Dims.make("float", img=2, chan=16, y=8, x=8))
    p.add(PipeOp("x", "Convolution", "data", "x", out_chans=16, kern_sz=(1, 1)))
    p.add(PipeOp("a", "Convolution", "x", "a", out_chans=8, kern_sz=(1, 1)))
    p.add(PipeOp("b", "Convolution", "x", "b", out_chans=8, kern_sz=(1, 1)))
    p.add(PipeOp("relu_x", "ReLU", "x", "x"))                       # in place on the shared bottom, AFTER a and b have read it
    p.add(PipeOp("c", "Convolution", "x", "c", out_chans=8, kern_sz=(1, 1)))
    p.add(PipeOp("d", "Convolution", "x", "d", out_chans=8, kern_sz=(1, 1)))
    m = [o for o in p.ops if o.tag in "abcd"]
    assert [[o.tag for o in r] for r in sibling_runs(p.ops, m)] == [["a", "b"], ["c", "d"]]
    assert [[o.tag for o in r] for r in sibling_runs(p.ops, m, fused={"relu_x"})] == [["a", "b", "c", "d"]]   # (a ReLU fused into its producer is not a separate writer)
    # an in-place op on ANOTHER node between the members changes nothing
    q = ConvPipe("t2", "data", Dims.make("float", img=2, chan=16, y=8, x=8))
    q.add(PipeOp("a", "Convolution", "data", "a", out_chans=8, kern_sz=(1, 1))); q.add(PipeOp("relu_a", "ReLU", "a", "a"))
    q.add(PipeOp("b", "Convolution", "data", "b", out_chans=8, kern_sz=(1, 1)))
    assert [[o.tag for o in r] for r in sibling_runs(q.ops, [q.ops[0], q.ops[2]])] == [["a", "b"]]
