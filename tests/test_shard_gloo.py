"""N>1 path on CPU: world_size=2 `gloo` process group.  Checks the sharding contract of boda_amd/shard.py -- contiguous
batch chunks, weights broadcast once from rank 0, no data-path collective, gathered shards == the unsharded result.
The per-rank arithmetic here is the CPU oracle standing in for the GPU kernels (this test is about the partitioning)."""
import os
import sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from boda_amd.shard import shard_range, shard_op
from boda_amd.op import parse_op, RtErr

SG = "(str_vals=(type=sgemm),nda_vals=(a=(dims=(K=96,M=70)),b=(dims=(K=96,N=40)),c=(dims=(M=70,N=40))))"
CV = ("(str_vals=(type=Convolution),nda_vals=(biases=(dims=(out_chan=12)),filts=(dims=(out_chan=12,in_chan=5,y=3,x=3)),"
      "in=(dims=(img=7,chan=5,y=9,x=9)),in_pad=(tn=none,dims=(y=1,x=1)),kern_sz=(tn=none,dims=(y=3,x=3)),"
      "out=(dims=(img=7,chan=12,y=9,x=9)),out_chans=(tn=uint32_t,v=12),stride=(tn=none,dims=(y=1,x=1))))")


def test_shard_ranges_partition_exactly():
    for total in (1, 7, 8, 256, 1000):
        for world in (1, 2, 3, 8):
            rs = [shard_range(total, r, world) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == total and all(a[1] == b[0] for a, b in zip(rs, rs[1:]))
            assert max(e - b for b, e in rs) - min(e - b for b, e in rs) <= 1
    with pytest.raises(RtErr):
        shard_op(parse_op(CV), 7, 8)  # 7 images over 8 ranks: last rank would own nothing
    o, b, tot = shard_op(parse_op(CV), 1, 2)
    assert (b, tot) == (4, 7) and o.get_dims("in").dsz("img") == 3 and o.get_dims("out").dsz("img") == 3
    assert o.get_dims("filts") == parse_op(CV).get_dims("filts")
    o, b, tot = shard_op(parse_op(SG), 0, 2)
    assert o.get_dims("a").dsz("M") == 35 and o.get_dims("c").dsz("M") == 35 and o.get_dims("b").dsz("N") == 40


def _worker(rank, world, port, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        sys.path.insert(0, ROOT)
        import torch
        import torch.distributed as dist
        from boda_amd.shard import shard_op, broadcast_weights, gather_outputs
        from boda_amd.op import parse_op
        from oracle import boda_oracle as bo
        dist.init_process_group("gloo", rank=rank, world_size=world)
        res = {}
        # ---- Convolution: shard on img
        op = parse_op(CV); g = op.conv_geom()
        lop, img0, B = shard_op(op, rank, world); lg = lop.conv_geom()
        full_in = bo.gen_conv_in(B, g["C"], g["H"], g["W"])
        loc_in = full_in[img0:img0 + lg["B"]]  # == what gen_data with ix_off=img0*C*H*W generates on the device
        if rank == 0:
            filts = bo.gen_conv_filts(g["OC"], g["C"], g["KH"], g["KW"]); biases = bo.gen_conv_biases(g["OC"])
        else:  # garbage until the broadcast
            filts = np.full((g["OC"], g["C"], g["KH"], g["KW"]), 1e9, np.float32); biases = np.full((g["OC"],), -1e9, np.float32)
        tf, tb = torch.from_numpy(filts), torch.from_numpy(biases)
        from boda_amd.shard import verify_replicated, tensor_digest
        from boda_amd.op import RtErr
        try:   # before the broadcast the ranks hold different bytes: the check must say so, on every rank
            verify_replicated([tf, tb]); res["caught_unequal"] = False
        except RtErr:
            res["caught_unequal"] = True
        broadcast_weights([tf, tb], src=0)
        res["verified"] = verify_replicated([tf, tb])          # ... and after it the same bytes everywhere
        res["digest_position_sensitive"] = tensor_digest(torch.tensor([1.0, 2.0])) != tensor_digest(torch.tensor([2.0, 1.0]))
        out = bo.conv_fwd(loc_in, tf.numpy(), tb.numpy(), (1, 1), (1, 1), True)
        full = gather_outputs(out, "Convolution", "out")
        if rank == 0:
            want = bo.conv_fwd(full_in, filts, biases, (1, 1), (1, 1), True)
            res["conv_equal"] = bool(np.array_equal(full, want))
        # ---- sgemm: shard on M (a is K:M -> columns)
        op = parse_op(SG); sg = op.sgemm_geom()
        lop, m0, M = shard_op(op, rank, world); lm = lop.sgemm_geom()["M"]
        a_full = bo.gen_sgemm_a(sg["K"], M); a_loc = np.ascontiguousarray(a_full[:, m0:m0 + lm])
        b = bo.gen_sgemm_b(sg["K"], sg["N"]) if rank == 0 else np.zeros((sg["K"], sg["N"]), np.float32)
        tbm = torch.from_numpy(b); broadcast_weights([tbm], 0)
        c = bo.sgemm(a_loc, tbm.numpy())
        cf = gather_outputs(c, "sgemm", "c")
        if rank == 0:
            res["sgemm_equal"] = bool(np.array_equal(cf, bo.sgemm(a_full, bo.gen_sgemm_b(sg["K"], sg["N"]))))
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, res, None))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, None, traceback.format_exc()))


def test_two_rank_gloo_sharded_equals_unsharded():
    import torch.multiprocessing as mp
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = [q.get(timeout=180) for _ in ps]
    for p in ps:
        p.join(timeout=60)
    for rank, res, err in got:
        assert err is None, err
    r0 = [res for rank, res, err in got if rank == 0][0]
    assert r0 == {"conv_equal": True, "sgemm_equal": True, "caught_unequal": True, "verified": True, "digest_position_sensitive": True}
    r1 = [res for rank, res, err in got if rank == 1][0]
    assert r1["caught_unequal"] and r1["verified"]      # (the comparison runs on every rank)
