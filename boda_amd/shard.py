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


def tensor_digest(t) -> int:
    """A 64-bit position-sensitive digest of a torch tensor's BYTES (any dtype / device): sum over its bytes, taken as little-endian 16-bit words w_i, of
    w_i * (i mod 65521 + 1), wrapped to int64.  Not cryptographic -- it tells a replicated weight tensor from one that was not replicated (zeros, another rank's
    pattern, a shifted copy)."""
    import torch
    b = t.contiguous().view(torch.uint8).reshape(-1)
    if b.numel() % 2:
        b = torch.cat([b, torch.zeros(1, dtype=torch.uint8, device=b.device)])
    w16 = b.view(torch.int16)
    total = 0
    CH = 1 << 22   # words per chunk: the int64 temporaries stay at ~100 MB whatever the tensor (the one-shot form took ~12x the tensor: 1.8 GB for AlexNet fc6)
    for o in range(0, w16.numel(), CH):
        w = w16[o:o + CH].to(torch.int64) & 0xffff
        idx = (torch.arange(o, o + w.numel(), dtype=torch.int64, device=w.device) % 65521) + 1
        total = (total + int((w * idx).sum().item())) & 0xffffffffffffffff
    return total - (1 << 64) if total >= (1 << 63) else total   # (wrapped to int64, as the one-shot sum)


def verify_replicated(tensors: Sequence, what: str = "weights") -> bool:
    """After broadcast_weights: every rank digests every tensor, the digests are all-gathered (8 bytes per tensor and rank) and compared ON EVERY RANK.  A mismatch
    raises RtErr naming the first tensor that differs and the ranks that disagree with rank 0; -> True when all ranks hold the same bytes (also with one rank)."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return True
    mine = torch.tensor([tensor_digest(t) for t in tensors], dtype=torch.int64)
    dev = tensors[0].device if (len(tensors) and dist.get_backend() == "nccl") else torch.device("cpu")
    mine = mine.to(dev)
    parts = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, mine)
    ref = parts[0].cpu()
    for r, p_ in enumerate(parts):
        bad = (p_.cpu() != ref).nonzero().reshape(-1)
        if bad.numel():
            raise RtErr(f"verify_replicated: {what} tensor #{int(bad[0])} on rank {r} differs from rank 0 after the broadcast (digest {int(p_.cpu()[bad[0]])} vs {int(ref[bad[0]])})")
    return True


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
