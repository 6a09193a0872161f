#!/usr/bin/env python3
"""Winograd F(2x2,3x3) path vs the direct kernel on 3x3 layers: time per call and mrd.  SHAPES="B:C:H:OC:pad,..." """
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from boda_amd.cnn_op import NATIVE_ARGS, OpTune, add_codegen_annotations
from boda_amd.op import Op, Dims, Nda
from boda_amd.digest import SsdsDiff
from boda_amd.rtc import make_rtc, RtcArg, RtcFuncCall, RtcFuncInfo
from boda_amd import gen_data as gd
def conv_op(B, C, H, OC, P, K=3, S=1):
    d = lambda n, s: Nda(Dims(n, s, "float")); none = lambda yx: Nda(Dims(("y", "x"), tuple(yx), "none"), "none")
    OH = (H + 2 * P - K) // S + 1
    return Op({"type": "Convolution"}, {"in": d(("img", "chan", "y", "x"), (B, C, H, H)), "filts": d(("out_chan", "in_chan", "y", "x"), (OC, C, K, K)),
               "biases": d(("out_chan",), (OC,)), "out": d(("img", "chan", "y", "x"), (B, OC, OH, OH)), "stride": none((S, S)), "in_pad": none((P, P)),
               "kern_sz": none((K, K)), "out_chans": Nda(None, "uint32_t", (OC,))})
rtc = make_rtc(); rtc.init(); rtc.compile(gd.func_infos())
shapes = [tuple(int(x) for x in s.split(":")) for s in os.environ.get("SHAPES", "256:256:13:384:1,256:384:13:384:1,256:384:13:256:1,64:64:56:64:1,64:128:28:128:1,64:256:14:256:1,64:512:7:512:1").split(",")]
for si, sh in enumerate(shapes):
    B, C, H, OC, P = sh[:5]
    op = conv_op(*sh)
    anno = add_codegen_annotations(op, OpTune(hip_dtype=os.environ.get("DTYPE", ""))); fn = anno.get_func_name(); g = f"{fn}__{si}"
    rtc.compile([RtcFuncInfo(g, "", [x for x, _ in NATIVE_ARGS[fn]], anno)])
    am = {}
    for an, io in NATIVE_ARGS[fn]:
        if io == "REF": am[an] = RtcArg.ref(anno.get_dims(an)); continue
        vn = f"{an}_{si}"; rtc.create_var_with_dims(vn, anno.get_dims(an)); am[an] = RtcArg.var(vn)
        if io == "IN": rtc.run(gd.gen_call("Convolution", an, vn, anno.get_dims(an), 5, 0.0))
    call = RtcFuncCall(g, am); fl = op.flops(); ref = None
    for algo in (("",) if os.environ.get("DIRECT_ONLY") else ("", "winograd_all")):
        rtc.set_tune("conv_algo", algo); rtc.set_var_to_zero(f"out_{si}")
        for _ in range(int(os.environ.get("SETTLE", "60"))): rtc.run(call)
        rtc.finish_and_sync(); rtc.release_per_call_id_data()
        import time
        rtc.finish_and_sync(); t0 = time.perf_counter()
        n = 20
        for _ in range(n): rtc.run(call)
        rtc.finish_and_sync(); ms = (time.perf_counter() - t0) * 1e3 / n; rtc.release_per_call_id_data()
        out = rtc.copy_var_to_nda(f"out_{si}")
        if ref is None: ref = out; note = ""
        else: note = "mrd vs direct " + "%.2e" % SsdsDiff.of(ref, out).mrd
        print(f"B{B} C{C} {H}x{H} OC{OC} p{P} algo={algo or 'direct':9s} {ms*1e3:8.1f} us  {fl/ms/1e9:6.1f} eff TF/s  {note}  {rtc.last_launch()['cfg']}", flush=True)
    rtc.set_tune("conv_algo", "")
    for an, io in NATIVE_ARGS[fn]:
        if io != "REF": rtc.release_var(f"{an}_{si}")
