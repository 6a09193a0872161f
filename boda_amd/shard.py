"""Batch-axis sharding of the hot path over the GPUs of one node (one process per GPU).

The reference is single-device (src/nvrtc_util.cc:189 uses device 0; there are no collectives anywhere in it).  The
path shards naturally: every Convolution is independent per image and sgemm is independent per row of c, so rank r of
`world` owns a contiguous chunk of `img` (sgemm: of `M`) of every IN/OUT tensor whose leading batch dim it is, while
the weights (Convolution filts/biases, sgemm b) are replicated.  Collectives on the data path: NONE.  The only exchange
is a one-time broadcast of the weights from rank 0 (RCCL over xGMI via torch.distributed, backend "nccl"; "gloo" in the
CPU tests) and, when a caller wants the whole output on one host, a gather of the per-rank host arrays.
"""
from __future__ import annotations
from typing import Dict, List, Sequence, Tuple

import numpy as np

from .op import Dims, Nda, Op, RtErr

BATCH_DIM = {"Convolution": ("img", ("in", "out")), "sgemm": ("M", ("a", "c"))}
WEIGHT_ARGS = {"Convolution": ("filts", "biases"), "sgemm": ("b",)}


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [begin, end) of `total` batch items owned by `rank`; the first total % world ranks get one extra."""
    if not (0 <= rank < world):
        raise RtErr(f"shard_range: rank {rank} not in [0,{world})")
    q, r = divmod(total, world)
    b = rank * q + min(rank, r)
    return b, b + q + (1 if rank < r else 0)


def shard_op(op: Op, rank: int, world: int) -> Tuple[Op, int, int]:
    """-> (this rank's op, first global batch index, global batch size).  Weight args keep their dims."""
    t = op.get_type()
    if t not in BATCH_DIM:
        raise RtErr(f"shard_op: op type {t!r} does not shard on the batch axis")
    dn, args = BATCH_DIM[t]
    total = op.get_dims(args[0]).dsz(dn)
    b, e = shard_range(total, rank, world)
    if e <= b:
        raise RtErr(f"shard_op: rank {rank}/{world} would own no {dn} items of {total}")
    o = op.copy()
    for an in args:
        d = op.get_dims(an)
        sizes = tuple((e - b) if n == dn else s for n, s in zip(d.names, d.sizes))
        o.nda_vals[an] = Nda(dims=Dims(d.names, sizes, d.tn), tn=d.tn)
    return o, b, total


def batch_axis(op_type: str, arg: str) -> int:
    """Axis of `arg` along which shards concatenate (a is K:M -> axis 1; c is M:N, in/out are img:... -> axis 0)."""
    return 1 if (op_type == "sgemm" and arg == "a") else 0


def broadcast_weights(tensors: Sequence, src: int = 0) -> None:
    """One-time weight broadcast (torch tensors, any device) from `src` over the default process group."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    for t in tensors:
        dist.broadcast(t, src=src)


def gather_outputs(local: np.ndarray, op_type: str, arg: str, dst: int = 0):
    """Host-side gather of per-rank output arrays into the global tensor on rank `dst` (None elsewhere)."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    parts: List = [None] * dist.get_world_size() if dist.get_rank() == dst else None
    dist.gather_object(np.ascontiguousarray(local), parts, dst=dst)
    if dist.get_rank() != dst:
        return None
    return np.concatenate(parts, axis=batch_axis(op_type, arg))
