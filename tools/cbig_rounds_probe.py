#!/usr/bin/env python3
"""Launch time of the staging-wave convolution kernel against the tile count (the convolution-side twin of tools/sgemm_rounds_probe.py): AlexNet conv3's geometry
(256 -> 384 channels, 3x3, 13x13 planes) at batch sizes that make 0.5 .. 4 rounds of 256 tiles, one tile form per run.   usage: python tools/cbig_rounds_probe.py [tile ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from boda_amd.cnn_op import OpTune, add_codegen_annotations
from boda_amd.ops_prof import OpsBackend, profile_rcg_call
from boda_amd.rtc import make_rtc
from tools.cbig_probe import conv_op
tiles = sys.argv[1:] or ["128x256x16x2x4x2x1x32x2x2", "64x512x16x1x8x2x1x32x2x2", "64x256x16x1x8x2x1x32x2x2", "128x128x16x2x2x2x1x32x2x2"]
os.environ["BODAHIP_CBIG_SPLIT"] = "off"
rtc = make_rtc(); rtc.init(); be = OpsBackend(rtc)
for tile in tiles:
    bi, bj = [int(x) for x in tile.split("x")[:2]]
    for want in (128, 256, 320, 384, 512, 640, 768, 896, 1024, 1280, 1536):
        ti = 384 // bi if 384 % bi == 0 else 384 // bi + 1
        tj = max(1, want // ti); B = (tj * bj) // 169          # the largest batch whose pels still fit tj tile columns
        tjr = -(-(B * 169) // bj); n = ti * tjr
        op = conv_op(B, 256, 13, 13, 384, 3, 3, 1, 1)
        anno = add_codegen_annotations(op, OpTune(hip_tile=tile))
        outs, prc = profile_rcg_call(be, anno, 5, 0.0, 8, tile=tile)
        best = min(prc.all_secs[2:])
        print(f"{tile} batch {B:4d}: {n:5d} tiles = {n / 256:5.2f} rounds of 256: {best * 1e6:8.1f} us = {best * 1e6 / (n / 256):7.1f} us per round-equivalent, {op.flops() / best / 1e12:6.1f} TF/s [{prc.launch['cfg']}]", flush=True)
