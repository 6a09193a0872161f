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
