#!/usr/bin/env python3
"""Reference point for the MFMA-bound bf16 layers: what the vendor GEMM (torch.matmul -> hipBLASLt) reaches on the plain-GEMM shapes of those layers on this box
(bf16 operands, fp32 accumulate, bf16 out).  AlexNet conv3 at 256 images as a GEMM is M = 384, N = 43264, K = 2304.  Measurement only: nothing in the product calls it."""
import sys, torch
shapes = [(384, 43264, 2304, "alexnet conv3 @256"), (384, 43264, 3456, "alexnet conv4 @256"), (256, 43264, 3456, "alexnet conv5 @256"), (256, 186624, 2400, "alexnet conv2 @256"),
          (512, 50176, 4608, "resnet res3 3x3 @64 (as GEMM)"), (8192, 8192, 8192, "8192^3"), (4096, 4096, 4096, "4096^3")]
dev = "cuda"
for M, N, K, name in shapes:
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16); b = torch.randn(K, N, device=dev, dtype=torch.bfloat16)
    for _ in range(10): c = a @ b
    torch.cuda.synchronize()
    for rep in range(2):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        n = 30
        ev[0].record()
        for _ in range(n): c = a @ b
        ev[1].record(); torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / n
    print(f"{name:32s} M{M} N{N} K{K}: {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:8.1f} TF/s", flush=True)
