#!/usr/bin/env python3
"""bench.py -- effective TFLOP/s of the native rtc_compute path on BASELINE.json's headline workload.

Workload at N=1 (config.workload): configs[1] = test/sgemm-ops-full.txt, the 17 square fp32 sgemms 64^3..12288^3, run
through be=hip's native door (hip_sgemm -> kernels/gemm_conv_f32.hip).  One "step" = one pass over all 17 ops
(8.66 TFLOP).  Inputs are the reference's deterministic gen_data mode-5 tensors, generated on the device and resident
in HBM before the timed region.  flops = 2*M*N*K (src/latex-util.H:116-120).
  --workload alexnet   BASELINE configs[2]: AlexNet-ng conv layers at batch 256 (hip_conv), same accounting.
N>1 (launched by torch.distributed.run, one process per GPU): the path shards on the batch axis with no data-path
collective -- every rank runs the same per-GPU workload on its own shard (sgemm: its own M-shard of an N-times-taller
problem; conv: its own images), weights (sgemm b / conv filts+biases) are broadcast once from rank 0 over RCCL before
the timed region.  value = all ranks' flops / max-over-ranks time ("scaling": "weak").
"""
from __future__ import annotations
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs @ 2.4 GHz
PEAK_HBM_GBS = 8000.0


ALEXNET_LAYERS = [  # alexnet_ng_conv: (in_chan, in_hw, out_chan, k, stride, pad) -- conv1..5, fc6..8
    (3, 227, 96, 11, 4, 0), (96, 27, 256, 5, 1, 2), (256, 13, 384, 3, 1, 1), (384, 13, 384, 3, 1, 1),
    (384, 13, 256, 3, 1, 1), (256, 6, 4096, 6, 1, 0), (4096, 1, 4096, 1, 1, 0), (4096, 1, 1000, 1, 1, 0)]
NIN_LAYERS = [  # nin_imagenet at 227x227: conv1 cccp1 cccp2 conv2 cccp3 cccp4 conv3 cccp5 cccp6 conv4 cccp7 cccp8
    (3, 227, 96, 11, 4, 0), (96, 55, 96, 1, 1, 0), (96, 55, 96, 1, 1, 0), (96, 27, 256, 5, 1, 2), (256, 27, 256, 1, 1, 0),
    (256, 27, 256, 1, 1, 0), (256, 13, 384, 3, 1, 1), (384, 13, 384, 1, 1, 0), (384, 13, 384, 1, 1, 0), (384, 6, 1024, 3, 1, 1),
    (1024, 6, 1024, 1, 1, 0), (1024, 6, 1000, 1, 1, 0)]


def conv_ops(layers, batch: int):
    from boda_amd.op import parse_op
    ops = []
    for C, H, OC, K, S, P in layers:
        O = (H + 2 * P - K) // S + 1
        ops.append(parse_op(f"(str_vals=(type=Convolution),nda_vals=(biases=(dims=(out_chan={OC})),filts=(dims=(out_chan={OC},in_chan={C},y={K},x={K})),"
                            f"in=(dims=(img={batch},chan={C},y={H},x={H})),in_pad=(tn=none,dims=(y={P},x={P})),kern_sz=(tn=none,dims=(y={K},x={K})),"
                            f"out=(dims=(img={batch},chan={OC},y={O},x={O})),out_chans=(tn=uint32_t,v={OC}),stride=(tn=none,dims=(y={S},x={S}))))"))
    return ops


def alexnet_b256_ops(batch: int = 256):
    return conv_ops(ALEXNET_LAYERS, batch)


def nin_ops(batch: int = 256):
    return conv_ops(NIN_LAYERS, batch)


def sgemm_full_ops():
    from boda_amd.op import read_ops
    return read_ops(os.path.join(ROOT, "tests", "golden", "ops", "sgemm-ops-full.txt"))


def cpu_baseline(workload: str, budget_s: float = 12.0) -> dict:
    """The CPU oracle (a port: the reference has no CPU path) timed on this host's cores on a bounded sample."""
    from oracle import boda_oracle as bo
    import numpy as np
    if workload == "sgemm-ops-full":
        sizes, flops, t_tot = [], 0.0, 0.0
        for n in (2048, 3072, 4096, 5120, 6144, 7168, 8192, 8192, 8192, 8192, 8192, 8192):  # sizes of the workload, ~budget_s of CPU work
            a = bo.gen_sgemm_a(n, n, 5); b = bo.gen_sgemm_b(n, n, 5)
            t = time.perf_counter(); bo.sgemm(a, b); dt = time.perf_counter() - t
            sizes.append(n); flops += 2.0 * n ** 3; t_tot += dt
            if t_tot > budget_s:
                break
        return {"value": flops / t_tot / 1e12, "unit": "TFLOP/s", "cores": bo.num_threads(), "kind": "port",
                "sample": f"oracle/boda_oracle.c bo_sgemm (OpenMP, fp32 fmaf) on sgemm-ops-full sizes {sizes}, {t_tot:.1f} s"}
    flops, t_tot, batch = 0.0, 0.0, 8
    layers = ALEXNET_LAYERS if workload == "alexnet" else NIN_LAYERS
    while t_tot < budget_s and batch <= 256:
        for op in conv_ops(layers, batch):
            g = op.conv_geom()
            i = bo.gen_conv_in(g["B"], g["C"], g["H"], g["W"]); f = bo.gen_conv_filts(g["OC"], g["C"], g["KH"], g["KW"]); b = bo.gen_conv_biases(g["OC"])
            t = time.perf_counter(); bo.conv_fwd(i, f, b, (g["SY"], g["SX"]), (g["PY"], g["PX"]), True); dt = time.perf_counter() - t
            flops += op.flops(); t_tot += dt
        batch *= 2
    return {"value": flops / t_tot / 1e12, "unit": "TFLOP/s", "cores": bo.num_threads(), "kind": "port",
            "sample": f"oracle/boda_oracle.c bo_conv_fwd (OpenMP) on the {workload} conv layers at batches 8..{batch//2}, {t_tot:.1f} s"}


def bench_full_net(a, rtc, rank, world, dist, torch) -> int:
    """BASELINE configs[3]: full-net forward (nin_imagenet / alexnet_ng_conv) through ConvPipeFwd, batch-sharded."""
    import numpy as np
    from boda_amd import gen_data as gd
    from boda_amd.conv_pipe import ConvPipeFwd, alexnet_ng_conv, nin_imagenet
    from boda_amd.shard import broadcast_weights
    cp = nin_imagenet(a.batch) if a.workload == "nin-net" else alexnet_ng_conv(a.batch)
    fwd = ConvPipeFwd(rtc)
    fwd.init(cp)  # params: deterministic pattern on device (rank 0's are broadcast below)
    d = cp.nodes["data"]
    rtc.run(gd.gen_call("Convolution", "in", "data", d, 5, 0.0, shard_off=rank * a.batch))
    rtc.finish_and_sync(); rtc.release_per_call_id_data()
    if dist is not None:
        broadcast_weights([rtc.torch_view(pn) for pn in fwd.op_param_names], src=0)
        torch.cuda.synchronize()
    for _ in range(a.warmup):
        fwd.run_fwd_device_only()
    if dist is not None:
        dist.barrier()
    import gc
    gc.collect(); gc.disable()  # a generation-2 collection (tens of ms with torch loaded) must not land between two launches
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dev_ms = []
    for _ in range(a.steps):
        ts = time.perf_counter(); dev_ms.append(fwd.run_fwd_device_only())
        if os.environ.get("BENCH_DEBUG"):
            print(f"step wall {1e3*(time.perf_counter()-ts):.3f} ms, device first-to-last {dev_ms[-1]:.3f} ms", file=sys.stderr)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX); elapsed = float(t.item())
    flops = float(cp.conv_flops())
    if rank == 0:
        conv_ms = sum(ms for _, f, ms, _ in fwd.per_call_ms if f == "hip_conv"); other_ms = sum(ms for _, f, ms, _ in fwd.per_call_ms if f != "hip_conv")
        value = flops * a.steps * world / elapsed / 1e12
        out = {"metric": "effective TFLOP/s (conv 2*M*N*K / whole-net forward time), whole job", "value": round(value, 3), "unit": "TFLOP/s",
               "per_gpu": round(value / world, 3), "images_per_s": round(a.batch * world * a.steps / elapsed, 1), "n_gpus": world, "steps": a.steps,
               "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f32", "data": "synthetic (reference gen_data mode 5 inputs and weights, generated on device)",
               "config": {"workload": f"{cp.name} full net forward (rtc_fwd counterpart), batch {a.batch}/GPU: {len(fwd.fwd_calls)} calls "
                                      f"({sum(1 for c in fwd.fwd_calls if c.func == 'hip_conv')} hip_conv + pool/lrn CUCL kernels)",
                          "parallelism": f"batch-shard x{world}, weights broadcast once (RCCL)", "device": rtc.get_plat_tag()},
               "roofline": {"bound": "mfma", "achieved": round(flops / (conv_ms * 1e-3) / 1e12, 3), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                            "frac": round(flops / (conv_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": None, "kernel": "bodahip_conv_f32",
                            "conv_ms": round(conv_ms, 4), "non_conv_ms": round(other_ms, 4), "device_ms_first_to_last_call": round(float(np.mean(dev_ms)), 4)},
               "per_call": [{"tag": t_, "func": f, "ms": round(ms, 5)} for t_, f, ms, _ in fwd.per_call_ms]}
        if a.per_op:
            for t_, f, ms, fl in fwd.per_call_ms:
                print(f"  {t_:8s} {f:10s} {ms:9.4f} ms" + (f" {fl/ms/1e9:8.2f} TF/s" if fl else ""), file=sys.stderr)
        print(json.dumps(out))
    fwd.release(); rtc.close()
    if dist is not None:
        dist.destroy_process_group()
    return 0


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="sgemm-ops-full", choices=["sgemm-ops-full", "alexnet", "nin", "nin-net", "alexnet-net"],
                    help="*-net: the whole network through the has_conv_fwd_t(mode=rtc) driver (convs + pool/LRN kernels), BASELINE configs[3]")
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch of the conv workloads")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--per-op", action="store_true", help="also print a per-op table to stderr")
    a = ap.parse_args()

    import numpy as np
    import torch
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            print(f"bench.py: --gpus {a.gpus} needs torch.distributed.run --nproc-per-node {a.gpus}", file=sys.stderr); return 2
    dist = None
    # test hook only: BENCH_SAME_GPU=1 runs all ranks on GPU 0 over gloo (exercises the N>1 code path on a 1-GPU box)
    same_gpu = os.environ.get("BENCH_SAME_GPU") == "1"
    if same_gpu:
        local_rank = 0
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if same_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))  # RCCL on ROCm
    torch.cuda.set_device(local_rank)

    from boda_amd import gen_data as gd
    from boda_amd.cnn_op import NATIVE_ARGS, OpTune, add_codegen_annotations
    from boda_amd.rtc import RtcArg, RtcFuncCall, RtcFuncInfo, make_rtc

    rtc = make_rtc("(be=hip)", local_rank)
    rtc.init()
    if a.workload.endswith("-net"):
        return bench_full_net(a, rtc, rank, world, dist, torch)
    rtc.compile(gd.func_infos()); rtc._gen_data_compiled = True
    ops = {"sgemm-ops-full": sgemm_full_ops, "alexnet": lambda: alexnet_b256_ops(a.batch), "nin": lambda: nin_ops(a.batch)}[a.workload]()
    from boda_amd.shard import WEIGHT_ARGS as weight_args, BATCH_DIM

    calls = []  # (op, RtcFuncCall)
    for i, op in enumerate(ops):
        anno = add_codegen_annotations(op, OpTune())
        fn = anno.get_func_name(); gen_fn = f"{fn}__{i}"
        rtc.compile([RtcFuncInfo(gen_fn, "", [x for x, _ in NATIVE_ARGS[fn]], anno)])
        am = {}
        for an, io in NATIVE_ARGS[fn]:
            if io == "REF":
                am[an] = RtcArg.ref(anno.get_dims(an)); continue
            vn = f"op{i}_{an}"
            rtc.create_var_with_dims(vn, anno.get_dims(an))
            am[an] = RtcArg.var(vn)
            if io == "IN":
                # rank r owns batch chunk r of a world-times-larger global problem: generate that slice of the global
                # pattern; weights are generated on rank 0 only and broadcast below (other ranks start from zeros)
                dn = BATCH_DIM[op.get_type()][0]
                is_w = an in weight_args[op.get_type()]
                if is_w and rank != 0:
                    continue
                per = anno.get_dims(an).dsz(dn) if (not is_w and anno.get_dims(an).has(dn)) else 0
                rtc.run(gd.gen_call(op.get_type(), an, vn, anno.get_dims(an), 5, 0.0, shard_off=rank * per, shard_glob=world * per))
        calls.append((op, RtcFuncCall(gen_fn, am)))
    rtc.finish_and_sync()
    if dist is not None:  # one-time weight broadcast from rank 0 over RCCL/xGMI (off the timed path)
        for (op, rfc) in calls:
            for an in weight_args[op.get_type()]:
                dist.broadcast(rtc.torch_view(rfc.arg_map[an].n), src=0)
        torch.cuda.synchronize()

    def step():
        return [rtc.run(rfc) for (_, rfc) in calls]

    for _ in range(a.warmup):
        step()
    rtc.finish_and_sync(); rtc.release_per_call_id_data()
    if dist is not None:
        dist.barrier()
    import gc
    gc.collect(); gc.disable()  # keep the collector out of the timed region (it is host-side noise, not work of the path)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ids = [step() for _ in range(a.steps)]
    rtc.finish_and_sync()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    if dist is not None:
        dist.barrier()
    elapsed = t1 - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX); elapsed = float(t.item())

    # per-kernel durations from the backend's HIP events (recorded on the stream the kernels run on)
    per_op_ms = np.array([[rtc.get_dur(i, i) for i in st] for st in ids])  # steps x ops
    flops_per_op = np.array([op.flops() for op, _ in calls], dtype=np.float64)
    bytes_per_op = np.array([op.algo_bytes() for op, _ in calls], dtype=np.float64)
    step_flops = float(flops_per_op.sum())
    kern_ms_per_step = float(per_op_ms.sum(axis=1).mean())
    value = step_flops * a.steps * world / elapsed / 1e12

    if rank == 0:
        avg_ms = per_op_ms.mean(axis=0)
        achieved = step_flops / (kern_ms_per_step * 1e-3) / 1e12
        li = rtc.last_launch(); info = rtc.get_device_info()
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_summary.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get(a.workload, {}).get("hbm_bytes_per_step")
            except Exception:
                traffic = None
        out = {
            "metric": "effective TFLOP/s (2*M*N*K / time), whole job", "value": round(value, 3), "unit": "TFLOP/s",
            "per_gpu": round(value / world, 3), "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic (reference gen_data mode 5, generated on device)",
            "config": {"workload": {"sgemm-ops-full": "test/sgemm-ops-full.txt: 17 fp32 sgemms 64^3..12288^3 via hip_sgemm",
                                    "alexnet": f"alexnet_ng_conv per-layer conv-ops, batch {a.batch}/GPU, via hip_conv",
                                    "nin": f"nin_imagenet per-layer conv-ops, batch {a.batch}/GPU, via hip_conv"}[a.workload],
                       "ops": len(calls), "tflop_per_step": round(step_flops / 1e12, 4), "parallelism": f"batch-shard x{world}, weights broadcast once (RCCL)",
                       "device": rtc.get_plat_tag(), "arch": info["arch"], "cus": info["num_cus"]},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 3), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": traffic,
                         "kernel": li["kernel"], "kernel_ms_per_step": round(kern_ms_per_step, 4),
                         "note": "achieved = sum(2MNK) / sum(HIP-event kernel durations) over the timed steps"},
            "per_op": [{"flops": float(f), "ms": round(float(m), 5), "tflops": round(float(f / (m * 1e-3) / 1e12), 2),
                        "gbs": round(float(b / (m * 1e-3) / 1e9), 1)} for f, b, m in zip(flops_per_op, bytes_per_op, avg_ms)],
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.workload)
        if a.per_op:
            for (op, _), r in zip(calls, out["per_op"]):
                print(f"  {op.get_type():12s} {r['flops']/1e9:10.2f} GF {r['ms']:9.4f} ms {r['tflops']:8.2f} TF/s {r['gbs']:9.1f} GB/s", file=sys.stderr)
        print(json.dumps(out))
    rtc.close()
    if dist is not None:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
