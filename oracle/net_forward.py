"""Layer-by-layer forward of a ConvPipe with the CPU oracle -- TEST INFRASTRUCTURE (like everything under oracle/): only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline may import it.  Restates the order of the reference's net-level forward
(src/rtc_fwd.cc:263-405: one op after the other in definition order, in-place ops on their bottom node)."""
from __future__ import annotations
from typing import Dict

import numpy as np

from . import boda_oracle as bo


def oracle_forward(cp, data: np.ndarray, params: Dict[str, np.ndarray], bf16: bool = False, store_bf16: bool = False) -> Dict[str, np.ndarray]:
    """Reference-order forward of `cp` with the CPU oracle.  bf16: conv operands rounded to bf16 (RNE) on the way in; store_bf16: every node
    is additionally STORED as bf16 (the channels-last bf16 nets: one more rounding per op output; implies bf16).
    Returns every node after its in-place ops (what the device vars hold after run_fwd)."""
    vals = {cp.in_node: bo.to_bf16(data) if store_bf16 else data}
    bf16 = bf16 or store_bf16
    for op in cp.ops:
        x = vals[op.bot]
        if op.type == "Convolution":
            f = params[op.tag + "_filts"]
            if bf16:   # what the bf16 kernels do: both operands rounded to bf16 (RNE) on the way in, fp32 accumulate, fp32 bias
                x, f = bo.to_bf16(x), bo.to_bf16(f)
            vals[op.top] = bo.conv_fwd(x, f, params[op.tag + "_biases"], op.stride, op.in_pad, relu=False)
        elif op.type == "ReLU":
            vals[op.top] = bo.relu(x)
        elif op.type == "Pooling":
            vals[op.top] = bo.pool_fwd(x, op.kern_sz, op.stride, op.in_pad, bool(op.avg_pool))
        elif op.type == "LRN":
            vals[op.top] = bo.lrn_fwd(x, *op.lrn)
        elif op.type == "Dropout":
            vals[op.top] = x
        elif op.type == "Concat":
            vals[op.top] = np.concatenate([vals[b] for b in op.bots], axis=1)
        if store_bf16:
            vals[op.top] = bo.to_bf16(vals[op.top])
    return vals
