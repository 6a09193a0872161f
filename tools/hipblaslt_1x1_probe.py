#!/usr/bin/env python3
"""Reference point for the channels-last bf16 1x1 convolutions of BASELINE config 5 (64 images): a 1x1 convolution on img:y:x:chan tensors IS the row-major GEMM
out[pels][out_chan] = in[pels][in_chan] @ W[in_chan][out_chan]; what the vendor GEMM (torch.matmul -> hipBLASLt; bf16 operands, fp32 accumulate, bf16 out, no bias / ReLU)
takes for it on this box, beside hip_conv_nhwc's own time for the layer (bias + ReLU included) in the same process.  Measurement only: nothing in the product calls it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from boda_amd.cnn_op import OpTune, add_codegen_annotations
from boda_amd.ops_prof import OpsBackend, profile_rcg_call
from boda_amd.rtc import make_rtc
rtc = make_rtc("(be=hip)", 0); rtc.init(); be = OpsBackend(rtc)
dev = "cuda"
for net in ("resnet-50", "googlenet_conv"):
    seen = {}
    for op in bench.net_conv_ops(net, 64):
        g = op.conv_geom()
        if g["KH"] == 1 and g["SY"] == 1 and g["OH"] > 1: seen.setdefault((g["C"], g["H"], g["OC"]), op)
    print(f"## {net} at 64 images: 1x1 / stride 1 layers -- C HxW OC | GEMM pels x K x OC | hip_conv_nhwc us (TB/s of its tensors) | vendor GEMM us (TB/s) | ours / vendor")
    for (C, H, OC), op in seen.items():
        pels = 64 * H * H
        anno = add_codegen_annotations(op, OpTune(hip_dtype="bf16", hip_layout="nhwc"))
        _, prc = profile_rcg_call(be, anno, 5, 0.0, 14, want_outs=False)
        ours = float(np.median(prc.all_secs[3:])) * 1e6
        a = torch.randn(pels, C, device=dev, dtype=torch.bfloat16); w = torch.randn(C, OC, device=dev, dtype=torch.bfloat16)
        for _ in range(10): c = a @ w
        torch.cuda.synchronize()
        ts = []
        for _ in range(14):     # one launch at a time between events, like the backend's per-call timing
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); c = a @ w; e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
        ven = float(np.median(ts[3:]))
        byt = 2.0 * (pels * C + pels * OC + C * OC)
        print(f"C{C:5d} {H:3d}x{H:<3d} OC{OC:5d} | {pels:6d} x {C:4d} x {OC:4d} | {ours:7.1f} us ({byt/ours/1e6:5.2f}) [{prc.launch['cfg']:>20s}] | {ven:7.1f} us ({byt/ven/1e6:5.2f}) | {ours/ven:5.2f}", flush=True)
        del a, w, c
rtc.close()
