"""-m gpu: N devices behind ONE rtc_compute_t (boda_amd/csrc/hip_multi.cc): vars with a leading `img` dim (sgemm: dim `M`) are
sharded, weights replicated, copy_nda_to_var scatters, run() enqueues on every device, copy_var_to_nda gathers.  The device list repeats GPU 0
({0,0}, {0,0,0}), so the sharding logic runs with the HIP kernels doing the arithmetic on a one-GPU box: gathered results must equal the
oracle -- and the single-device backend -- bit for bit, for even, uneven and empty shards."""
import os
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from boda_amd.cnn_op import NATIVE_ARGS, OpTune, add_codegen_annotations
from boda_amd.op import Dims, Op, RtErr, UnsupErr, parse_op
from boda_amd.rtc import RtcArg, RtcFuncCall, RtcFuncInfo, make_rtc
from oracle import boda_oracle as bo


def _run(rtc, op, ins, tune=None):
    anno = add_codegen_annotations(op, tune or OpTune()); fn = anno.get_func_name()
    rtc.compile([RtcFuncInfo("f", "", [a for a, _ in NATIVE_ARGS[fn]], anno)])
    am, made = {}, []
    try:
        for an, io in NATIVE_ARGS[fn]:
            if io == "REF":
                am[an] = RtcArg.ref(anno.get_dims(an)); continue
            rtc.create_var_with_dims(an, anno.get_dims(an)); made.append(an); am[an] = RtcArg.var(an)
            assert rtc.get_var_dims(an) == anno.get_dims(an)       # the caller sees the logical dims
            if io == "IN":
                rtc.copy_nda_to_var(an, ins[an])
        ids = [rtc.run(RtcFuncCall("f", am)) for _ in range(2)]
        rtc.finish_and_sync()
        assert rtc.get_dur(ids[0], ids[1]) > 0
        outs = {an: rtc.copy_var_to_nda(an) for an, io in NATIVE_ARGS[fn] if io != "REF"}
        return outs
    finally:
        for vn in made:
            rtc.release_var(vn)
        rtc.release_func("f"); rtc.release_per_call_id_data()


def _conv_op(B, C, H, W, OC, KH, KW, S, P):
    OH = (H + 2 * P - KH) // S + 1; OW = (W + 2 * P - KW) // S + 1
    return parse_op(f"(str_vals=(type=Convolution),nda_vals=(biases=(dims=(out_chan={OC})),filts=(dims=(out_chan={OC},in_chan={C},y={KH},x={KW})),"
                    f"in=(dims=(img={B},chan={C},y={H},x={W})),in_pad=(tn=none,dims=(y={P},x={P})),kern_sz=(tn=none,dims=(y={KH},x={KW})),"
                    f"out=(dims=(img={B},chan={OC},y={OH},x={OW})),out_chans=(tn=uint32_t,v={OC}),stride=(tn=none,dims=(y={S},x={S}))))")


@pytest.fixture(scope="module", params=[2, 3])
def multi(request):
    r = make_rtc("(be=hip,devices=" + ":".join(["0"] * request.param) + ")")
    r.init()
    assert r.get_plat_tag().endswith(f"*{request.param}")
    yield r, request.param
    r.finish_and_sync(); r.close()


@pytest.fixture(scope="module")
def single():
    r = make_rtc("(be=hip)", 0); r.init()
    yield r
    r.finish_and_sync(); r.close()


@pytest.mark.parametrize("shape", [(5, 32, 14, 14, 64, 5, 5, 1, 2), (1, 19, 11, 11, 40, 1, 1, 1, 0), (8, 3, 35, 35, 96, 11, 11, 4, 0), (7, 64, 9, 9, 130, 3, 3, 1, 1), (2, 256, 6, 6, 512, 6, 6, 1, 0)])
def test_conv_sharded_over_devices_equals_unsharded(multi, single, shape):
    rtc, n = multi
    B, C, H, W, OC, KH, KW, S, P = shape
    ins = {"in": bo.gen_conv_in(B, C, H, W), "filts": bo.gen_conv_filts(OC, C, KH, KW), "biases": bo.gen_conv_biases(OC)}
    got = _run(rtc, _conv_op(*shape), ins)
    want = bo.conv_fwd(ins["in"], ins["filts"], ins["biases"], (S, S), (P, P), True)
    assert np.array_equal(got["out"], want)                                     # gathered shards == the oracle's unsharded result
    assert np.array_equal(got["out"], _run(single, _conv_op(*shape), ins)["out"])   # == the single-device backend
    for an in ("in", "filts", "biases"):                                        # scatter / broadcast followed by gather gives the inputs back
        assert np.array_equal(got[an], ins[an]), an


@pytest.mark.parametrize("M,N,K", [(100, 36, 50), (1, 64, 64), (257, 130, 70), (512, 512, 96)])
def test_sgemm_sharded_on_M(multi, single, M, N, K):
    """a is K:M (sharded along its SECOND dim: packed per device), b K:N replicated, c M:N sharded along its leading dim."""
    rtc, n = multi
    op = parse_op(f"(str_vals=(type=sgemm),nda_vals=(a=(dims=(K={K},M={M})),b=(dims=(K={K},N={N})),c=(dims=(M={M},N={N}))))")
    ins = {"a": bo.gen_sgemm_a(K, M), "b": bo.gen_sgemm_b(K, N)}
    got = _run(rtc, op, ins)
    assert np.array_equal(got["c"], bo.sgemm(ins["a"], ins["b"]))
    assert np.array_equal(got["a"], ins["a"]) and np.array_equal(got["b"], ins["b"])


def test_multi_device_contract(multi):
    rtc, n = multi
    # generated CUCL source: runs (on every device) on replicated vars, refused on sharded ones
    src = "CUCL_GLOBAL_KERNEL void add1( GASQ float * const a, uint32_t const n ) { if( GLOB_ID_1D < n ) { a[GLOB_ID_1D] += 1.0f; } }\n"
    rtc.compile([RtcFuncInfo("add1", src, ["a", "n"], Op({"type": "x", "func_name": "add1"}, {}))])
    rtc.create_var_with_dims("r", Dims(("v",), (1000,), "float"))
    rtc.create_var_with_dims("s", Dims(("img", "chan"), (10, 100), "float"))
    try:
        rtc.copy_nda_to_var("r", np.arange(1000, dtype=np.float32))
        rtc.run(RtcFuncCall("add1", {"a": RtcArg.var("r"), "n": RtcArg.scalar(1000, "uint32_t")}, tpb=256, blks=4))
        rtc.finish_and_sync()
        assert np.array_equal(rtc.copy_var_to_nda("r"), np.arange(1000, dtype=np.float32) + 1)
        with pytest.raises(UnsupErr):
            rtc.run(RtcFuncCall("add1", {"a": RtcArg.var("s"), "n": RtcArg.scalar(1000, "uint32_t")}, tpb=256, blks=4))
        with pytest.raises(RtErr):
            rtc.get_var_raw_native_pointer("s")            # a sharded var has no single device pointer
        assert rtc.get_var_raw_native_pointer("r") != 0
        x = np.arange(1000, dtype=np.float32).reshape(10, 100)
        rtc.copy_nda_to_var("s", x)
        assert np.array_equal(rtc.copy_var_to_nda("s"), x)
        rtc.set_var_to_zero("s")
        assert not rtc.copy_var_to_nda("s").any()
    finally:
        rtc.release_var("r"); rtc.release_var("s"); rtc.release_func("add1"); rtc.release_per_call_id_data()


# ---------------------------------------------------------------------------------------------------------------
# generated (CUCL-source) functions on sharded vars: per-element functions that declare `// CUCL IX GLOB_ID_1D <arg>` run on each device over
# the ids of its own images (whole-tensor indices, shard pointers moved back by the shard's offset) -- csrc/hip_multi.cc
# ---------------------------------------------------------------------------------------------------------------
def test_generated_per_element_function_on_sharded_vars(multi):
    rtc, n = multi
    decl = "CUCL_GLOBAL_KERNEL void {name}( GASQ float * const a, GASQ float const * const b, uint32_t const n ) {{\n{ix}  if( GLOB_ID_1D >= n ) {{ return; }}\n{body}}}\n"
    src = (decl.format(name="addix", ix="  // CUCL IX GLOB_ID_1D a\n", body="  a[GLOB_ID_1D] += b[GLOB_ID_1D / 100] + (float)GLOB_ID_1D;\n")          # NOT idempotent: every id exactly once
           + decl.format(name="noix", ix="", body="  a[GLOB_ID_1D] += 1.0f;\n")
           + decl.format(name="grp", ix="  // CUCL IX GLOB_ID_1D a\n", body="  a[GLOB_ID_1D] += (float)LOC_ID_1D;\n"))
    op = lambda f: Op({"type": "x", "func_name": f}, {})
    rtc.compile([RtcFuncInfo("addix", src, ["a", "b", "n"], op("addix")), RtcFuncInfo("noix", "", ["a", "b", "n"], op("noix")), RtcFuncInfo("grp", "", ["a", "b", "n"], op("grp"))])
    for B in (10, 1, n - 1 if n > 1 else 1):       # incl. batches smaller than the device count (empty shards)
        rtc.create_var_with_dims("s", Dims(("img", "chan"), (B, 100), "float")); rtc.create_var_with_dims("t", Dims(("img",), (B,), "float"))
        try:
            x = np.arange(B * 100, dtype=np.float32).reshape(B, 100) * np.float32(0.5); y = np.arange(B, dtype=np.float32) * np.float32(1000.0)
            rtc.copy_nda_to_var("s", x); rtc.copy_nda_to_var("t", y)
            am = {"a": RtcArg.var("s"), "b": RtcArg.var("t"), "n": RtcArg.scalar(B * 100, "uint32_t")}
            cid = rtc.run(RtcFuncCall("addix", am, tpb=64, blks=(B * 100 + 63) // 64)); rtc.finish_and_sync()
            assert rtc.get_dur(cid, cid) >= 0
            assert np.array_equal(rtc.copy_var_to_nda("s"), x + y[:, None] + np.arange(B * 100, dtype=np.float32).reshape(B, 100))
            for bad in ("noix", "grp"):            # no index declaration / workgroup-level function: refused on sharded vars, as before
                with pytest.raises(UnsupErr):
                    rtc.run(RtcFuncCall(bad, am, tpb=64, blks=(B * 100 + 63) // 64))
        finally:
            rtc.release_var("s"); rtc.release_var("t"); rtc.release_per_call_id_data()
    for f in ("addix", "noix", "grp"):
        rtc.release_func(f)


def test_gen_data_on_a_sharded_var_gives_the_global_pattern(multi):
    """The reference's flat-index hash pattern (test/rtc/gen_data_Convolution_in.cucl: det_hash_rand( GLOB_ID_1D + c )) generated ON the devices,
    each over the ids of its own images: the gathered tensor is the whole-tensor pattern."""
    from boda_amd import gen_data as gd
    rtc, n = multi
    if not getattr(rtc, "_gen_data_compiled", False):
        rtc.compile(gd.func_infos()); rtc._gen_data_compiled = True
    d = Dims(("img", "chan", "y", "x"), (7, 5, 9, 11), "float")
    rtc.create_var_with_dims("gin", d)
    try:
        rtc.run(gd.gen_call("Convolution", "in", "gin", d, 5, 0.0)); rtc.finish_and_sync()
        assert np.array_equal(rtc.copy_var_to_nda("gin"), bo.gen_conv_in(7, 5, 9, 11))
    finally:
        rtc.release_var("gin"); rtc.release_per_call_id_data()


def test_gen_data_sgemm_a_on_a_var_sharded_along_its_second_dim(multi):
    """sgemm `a` is K:M and shards along M (every device holds K x M_i packed).  gen_data_sgemm_a declares what it walks (`// CUCL SHARD2 a size=M off=m_off`), so the
    backend runs it per device with the shard's extent and first column: the gathered tensor is the whole-tensor pattern (modes 5 and 600), and an sgemm on the
    device-generated operands equals the oracle's (test/rtc/gen_data_sgemm_a.cucl; round-4 gap: refused with unsup_err)."""
    from boda_amd import gen_data as gd
    rtc, n = multi
    if not getattr(rtc, "_gen_data_compiled", False):
        rtc.compile(gd.func_infos()); rtc._gen_data_compiled = True
    K, M = 37, 4 * n + 3          # (uneven shards; with n devices some hold one column more)
    d = Dims(("K", "M"), (K, M), "float")
    rtc.create_var_with_dims("ga", d)
    try:
        for mode in (5, 600):
            rtc.run(gd.gen_call("sgemm", "a", "ga", d, mode, 0.0)); rtc.finish_and_sync()
            assert np.array_equal(rtc.copy_var_to_nda("ga"), bo.gen_sgemm_a(K, M, mode)), mode
        # as one rank's M-shard of a larger global problem (bench.py's multi-process form on top of a multi-device backend): offsets add up
        rtc.run(gd.gen_call("sgemm", "a", "ga", d, 5, 0.0, shard_off=M, shard_glob=3 * M)); rtc.finish_and_sync()
        assert np.array_equal(rtc.copy_var_to_nda("ga"), bo.gen_sgemm_a(K, 3 * M, 5)[:, M:2 * M])
    finally:
        rtc.release_var("ga"); rtc.release_per_call_id_data()


@pytest.mark.parametrize("net,batch,ndev", [("nin", 16, 4), ("alexnet", 5, 3), ("nin-chain", 7, 3)])
def test_full_net_forward_on_a_multi_device_backend_equals_single_device(single, net, batch, ndev):
    """BASELINE config 4 behind the boundary: the whole net through ConvPipeFwd on (be=hip,devices=0:0:..) -- inputs and weights generated on the
    devices, hip_conv on every shard, the templated pool / LRN kernels over each shard's ids -- node for node bit-identical to one device.
    nin-chain: with cccp1 -> cccp2 as one hip_conv_k1_chain call per shard (the size gate lifted), against the UNFUSED pass on one device."""
    from boda_amd import gen_data as gd
    from boda_amd.conv_pipe import ConvPipeFwd as _CPF, alexnet_ng_conv, nin_imagenet
    chain = net == "nin-chain"; net = net.split("-")[0]
    ConvPipeFwd = (lambda r: _CPF(r, fuse_k1_chains=("all" if (chain and r is not single) else False))) if chain else _CPF
    cp_of = {"nin": nin_imagenet, "alexnet": alexnet_ng_conv}[net]
    res = []
    for be in ("(be=hip,devices=" + ":".join(["0"] * ndev) + ")", None):
        rtc = make_rtc(be) if be else single
        if be:
            rtc.init()
        cp = cp_of(batch)
        fwd = ConvPipeFwd(rtc); fwd.init(cp)
        try:
            rtc.run(gd.gen_call("Convolution", "in", fwd.in_var, cp.nodes["data"], 5, 0.0)); rtc.finish_and_sync()
            nodes = [nn for nn in cp.nodes if nn in {o.top for o in cp.ops if o.type != "Dropout"}]
            if chain:
                assert (fwd.k1_chains == [("cccp1", "cccp2")]) == (rtc is not single)
                nodes = [nn for nn in nodes if nn != "cccp1"]     # (the fused pass does not write it)
            for c in fwd.fwd_calls:
                c.call_id = rtc.run(c.rfc)
            rtc.finish_and_sync()
            assert rtc.get_dur(fwd.fwd_calls[0].call_id, fwd.fwd_calls[-1].call_id) > 0
            res.append({nn: rtc.copy_var_to_nda(fwd.var_of(nn)) for nn in ["data"] + nodes})
        finally:
            fwd.release(); rtc.release_per_call_id_data()
            if be:
                for k in ("_pool_funcs",):
                    rtc.__dict__.pop(k, None)
                rtc.close()
    assert np.array_equal(res[0]["data"], bo.gen_conv_in(*cp.nodes["data"].sizes))
    for nn in res[1]:
        assert np.array_equal(res[0][nn], res[1][nn]), nn
    assert np.abs(res[1][cp.out_node()]).max() > 0


@pytest.mark.parametrize("net,batch,ndev", [("googlenet", 7, 3), ("alexnet", 5, 2)])
def test_channels_last_net_and_graph_replay_on_a_multi_device_backend(single, net, batch, ndev, monkeypatch):
    """BASELINE config 5's form behind the boundary: a channels-last bf16 net through ConvPipeFwd on (be=hip,devices=0:0:..) -- the layout pass, pool / LRN / Concat
    kernels over each shard's ids (they walk 16-byte chunks: `// CUCL IX GLOB_ID_1D <arg> n=<count>`; the LRN kernels are `wave_local`), sibling groups, level sets
    and fused poolings on every shard -- node for node bit-identical to one device; then the whole pass captured into one hipGraph PER DEVICE and replayed as one call."""
    from boda_amd import gen_data as gd
    from boda_amd.cnn_op import OpTune
    from boda_amd.conv_pipe import ConvPipeFwd, alexnet_ng_conv, googlenet_conv
    monkeypatch.setenv("BODAHIP_NO_NHWC_SPLITK", "1")      # (a shard's tile count differs from the whole batch's: keep the planner from slicing K on one side only)
    cp_of = {"googlenet": googlenet_conv, "alexnet": alexnet_ng_conv}[net]
    res = []
    for be in ("(be=hip,devices=" + ":".join(["0"] * ndev) + ")", None):
        rtc = make_rtc(be) if be else single
        if be:
            rtc.init()
        cp = cp_of(batch)
        fwd = ConvPipeFwd(rtc, OpTune(hip_dtype="bf16", hip_layout="nhwc"), fuse_pools=True); fwd.init(cp)
        try:
            rtc.run(gd.gen_call("Convolution", "in", fwd.in_var, cp.nodes["data"], 5, 0.0)); rtc.finish_and_sync()
            nodes = [nn for nn in cp.nodes if nn != "data" and nn in {o.top for o in cp.ops if o.type != "Dropout"}]
            io = {}
            fwd.run_fwd([], io, nodes)
            res.append(io)
            if be:
                out = cp.out_node()
                n = fwd.capture_graph(); assert n == len(fwd.fwd_calls)
                for _ in range(2):
                    rtc.set_var_to_zero(fwd.var_of(out)); ms = fwd.run_graph(); assert ms > 0
                    assert np.array_equal(fwd._fetch(out), io[out])
                if net == "googlenet":
                    assert len(fwd.level_sets) >= 9 and len(fwd.fused_pools) == 9 and len(fwd.groups) == 9
                # round 5: the LRN -> Pooling pass through LDS -- a WORKGROUP function -- runs on the shards too (`// CUCL IX GRP_ID_1D in n=n`: whole workgroups per image)
                assert sum(c.func == "nhwc_lrn_pool_lds" for c in fwd.fwd_calls) == {"googlenet": 1, "alexnet": 2}[net]
        finally:
            fwd.release(); rtc.release_per_call_id_data()
            if be:
                rtc.close()
    for nn in res[1]:
        assert np.array_equal(res[0][nn], res[1][nn]), nn
    assert np.abs(res[1][cp.out_node()]).max() > 0


def test_ops_prof_end_to_end_on_a_sharded_backend(golden_dir):
    """The ops-prof protocol (src/rtc_prof.cc:44-126,194-371: vars, inputs generated ON DEVICE, run, read back, digests vs the reference's wisdom
    file) through ONE backend over three shards: the reference-held digests of test/good_tr/conv-debug are met by the gathered outputs."""
    import io
    from boda_amd.ops_prof import ops_prof
    from boda_amd.op import read_ops
    from boda_amd.digest import read_wisdoms
    rtc = make_rtc("(be=hip,devices=0:0:0)"); rtc.init()
    try:
        ops = read_ops(os.path.join(golden_dir, "ops", "conv-ops-debug-tmp.txt"))
        win = read_wisdoms(os.path.join(golden_dir, "wisdom", "conv-debug.wis"))
        buf = io.StringIO()
        wout, nfail, rows = ops_prof(rtc, ops, {"def": OpTune(), "t64": OpTune(hip_tile="64x64x16x1x1")}, "def", 5, wisdom_in=win, write_runs=True, out=buf)
        assert nfail == 0 and "***ALL IS WELL***" in buf.getvalue(), buf.getvalue()
        for w, wi in zip(wout, win):
            assert w.kgs[0][1].mrd_comp(wi.kgs[0][1], 2e-4) == ""
    finally:
        rtc.close()


def test_uneven_eight_way_split_at_config4_size(single):
    """BASELINE config 4's first layer at (nearly) its full batch: NiN conv1 on 1021 images (in 631 MB, out 1.18 GB: over 2 GiB / 2 as ONE tensor, every
    shard far below the 2 GiB-per-launch limit) split 8 ways unevenly (127 / 128 images), data generated on the devices.  Gathered result ==
    the single-device result bit for bit; first two and last image == the oracle (batch-prefix invariance)."""
    from boda_amd import gen_data as gd
    B = 1021
    op = _conv_op(B, 3, 227, 227, 96, 11, 11, 4, 0)
    anno = add_codegen_annotations(op, OpTune()); fn = anno.get_func_name()
    outs = []
    for be in ("(be=hip,devices=0:0:0:0:0:0:0:0)", None):
        rtc = make_rtc(be) if be else single
        if be:
            rtc.init()
        if not getattr(rtc, "_gen_data_compiled", False):
            rtc.compile(gd.func_infos()); rtc._gen_data_compiled = True
        rtc.compile([RtcFuncInfo("big", "", [a for a, _ in NATIVE_ARGS[fn]], anno)])
        am, made = {}, []
        try:
            for an, io_ in NATIVE_ARGS[fn]:
                if io_ == "REF":
                    am[an] = RtcArg.ref(anno.get_dims(an)); continue
                rtc.create_var_with_dims("big_" + an, anno.get_dims(an)); made.append("big_" + an); am[an] = RtcArg.var("big_" + an)
                if io_ == "IN":
                    rtc.run(gd.gen_call("Convolution", an, "big_" + an, anno.get_dims(an), 5, 0.0))
            rtc.run(RtcFuncCall("big", am)); rtc.finish_and_sync()
            outs.append(rtc.copy_var_to_nda("big_out"))
        finally:
            for vn in made:
                rtc.release_var(vn)
            rtc.release_func("big"); rtc.release_per_call_id_data()
            if be:
                rtc.close()
    assert np.array_equal(outs[0], outs[1])
    f = bo.gen_conv_filts(96, 3, 11, 11); bi = bo.gen_conv_biases(96)
    full_in_head = bo.gen_conv_in(B, 3, 227, 227)[[0, 1, B - 1]]
    want = bo.conv_fwd(full_in_head, f, bi, (4, 4), (0, 0), True)
    assert np.array_equal(outs[0][[0, 1, B - 1]], want)


def test_peer_fan_out_of_replicated_vars_is_ordered(monkeypatch):
    """copy_nda_to_var of a replicated var = one H2D to device 0 + device-to-device copies to the others, read on THEIR streams: device 0's stream
    must wait for those reads before it overwrites the buffer.  BODAHIP_FORCE_PEER=1 takes the device-to-device path between shards of one GPU
    (two distinct GPUs take it by themselves); back-to-back uploads into the same var must leave every replica holding the LAST upload --
    checked through a convolution, which reads each device's own replica."""
    monkeypatch.setenv("BODAHIP_FORCE_PEER", "1")
    rtc = make_rtc("(be=hip,devices=0:0:0)"); rtc.init()
    try:
        shape = (6, 64, 28, 28, 512, 3, 3, 1, 1)
        B, C, H, W, OC, KH, KW, S, P = shape
        ins = {"in": bo.gen_conv_in(B, C, H, W), "filts": bo.gen_conv_filts(OC, C, KH, KW), "biases": bo.gen_conv_biases(OC)}
        op = _conv_op(*shape); anno = add_codegen_annotations(op, OpTune()); fn = anno.get_func_name()
        rtc.compile([RtcFuncInfo("f", "", [a for a, _ in NATIVE_ARGS[fn]], anno)])
        am = {}
        for an, io_ in NATIVE_ARGS[fn]:
            if io_ == "REF":
                am[an] = RtcArg.ref(anno.get_dims(an)); continue
            rtc.create_var_with_dims(an, anno.get_dims(an)); am[an] = RtcArg.var(an)
        rtc.copy_nda_to_var("in", ins["in"]); rtc.copy_nda_to_var("biases", ins["biases"])
        for k in range(4):     # uploads chase each other: junk, junk, junk, the real filters
            rtc.copy_nda_to_var("filts", ins["filts"] if k == 3 else np.full_like(ins["filts"], float(k + 1)))
        rtc.run(RtcFuncCall("f", am)); rtc.finish_and_sync()
        assert np.array_equal(rtc.copy_var_to_nda("out"), bo.conv_fwd(ins["in"], ins["filts"], ins["biases"], (S, S), (P, P), True))
        rtc.set_var_to_zero("filts"); rtc.run(RtcFuncCall("f", am)); rtc.finish_and_sync()      # every replica zeroed: out = relu(bias)
        assert np.array_equal(rtc.copy_var_to_nda("out"), np.broadcast_to(np.maximum(ins["biases"], 0)[None, :, None, None], (B, OC, H, W)))
    finally:
        rtc.close()


def test_peer_fan_out_between_two_distinct_gpus():
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (the one-GPU box covers the same code path with BODAHIP_FORCE_PEER)")
    rtc = make_rtc("(be=hip,devices=0:1)"); rtc.init()
    try:
        shape = (6, 64, 28, 28, 512, 3, 3, 1, 1)
        B, C, H, W, OC, KH, KW, S, P = shape
        ins = {"in": bo.gen_conv_in(B, C, H, W), "filts": bo.gen_conv_filts(OC, C, KH, KW), "biases": bo.gen_conv_biases(OC)}
        got = _run(rtc, _conv_op(*shape), ins)
        assert np.array_equal(got["out"], bo.conv_fwd(ins["in"], ins["filts"], ins["biases"], (S, S), (P, P), True))
    finally:
        rtc.close()
