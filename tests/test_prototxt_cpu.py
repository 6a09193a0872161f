"""prototxt reader + shape inference (boda_amd/prototxt.py) and the GoogLeNet / ResNet-50 conv-op fixtures built with it."""
import os
from boda_amd.prototxt import parse, conv_ops
from boda_amd.op import data_path, read_ops

TXT = """
name: "tiny"  # comment
layer { name: "data" type: "Data" top: "data" top: "label" include { phase: TEST } transform_param { crop_size: 32 } }
layer { name: "data" type: "Data" top: "data" top: "label" include { phase: TRAIN } transform_param { crop_size: 99 } }
layer { name: "c1" type: "Convolution" bottom: "data" top: "c1" convolution_param { num_output: 8 pad: 1 kernel_size: 3 stride: 2 } }
layer { name: "r1" type: "ReLU" bottom: "c1" top: "c1" }
layer { name: "p1" type: "Pooling" bottom: "c1" top: "p1" pooling_param { pool: MAX kernel_size: 3 stride: 2 } }
layer { name: "b1" type: "Convolution" bottom: "p1" top: "b1" convolution_param { num_output: 4 kernel_size: 1 } }
layer { name: "b2" type: "Convolution" bottom: "p1" top: "b2" convolution_param { num_output: 6 kernel_size: 3 pad: 1 } }
layer { name: "cat" type: "Concat" bottom: "b1" bottom: "b2" top: "cat" }
layer { name: "gp" type: "Pooling" bottom: "cat" top: "gp" pooling_param { pool: AVE global_pooling: true } }
layer { name: "fc" type: "InnerProduct" bottom: "gp" top: "fc" inner_product_param { num_output: 10 } }
layer { name: "loss" type: "SoftmaxWithLoss" bottom: "fc" bottom: "label" top: "loss" }
"""


def test_parse_and_shape_inference():
    root = parse(TXT)
    assert len(root["layer"]) == 11 and root["name"] == ["tiny"]
    ops = dict(conv_ops(TXT, 5))
    assert list(ops) == ["c1", "b1", "b2", "fc"]
    assert ops["c1"].get_dims("in").sizes == (5, 3, 32, 32) and ops["c1"].get_dims("out").sizes == (5, 8, 16, 16)
    assert ops["b1"].get_dims("in").sizes == (5, 8, 8, 8)           # pool: ceil((16-3)/2)+1 = 8
    assert ops["b2"].get_dims("out").sizes == (5, 6, 8, 8)
    assert ops["fc"].get_dims("filts").sizes == (10, 10, 1, 1)      # concat 4+6 channels, global pool -> 1x1, IP as conv


def test_net_fixtures_consistent(golden_dir):
    g = read_ops(data_path("ops", "googlenet_conv-conv-ops-b1.txt"))
    r = read_ops(data_path("ops", "resnet-50-conv-ops-b1.txt"))
    assert len(g) == 64 and len(r) == 54
    key = lambda o: tuple(o.conv_geom()[k] for k in ("C", "H", "OC", "KH", "SY", "PY"))
    ref_shapes = {key(o) for o in read_ops(os.path.join(golden_dir, "ops", "conv-ops-1-5-20-nin-alex-gn.txt")) if o.conv_geom()["B"] == 1}
    assert all(key(o) in ref_shapes for o in g)   # every GoogLeNet conv shape we infer is in the reference's own op list
    assert len({key(o) for o in r}) == 21          # unique shapes incl. the fc (two 1x1 shapes coincide under this key)
    assert abs(sum(o.flops() for o in r) / 1e9 - 7.716) < 0.01
