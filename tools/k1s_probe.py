#!/usr/bin/env python3
"""Streaming 1x1 kernel (k1_stream_f32.hip) vs the tiled kernel: per-launch time and bit-equality of the outputs.
   SHAPES="B:C:H:OC,..."  SPECS="off,auto,1x4x3x2,..."  (auto = empty spec)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from boda_amd.cnn_op import NATIVE_ARGS, OpTune, add_codegen_annotations
from boda_amd.op import Op, Dims, Nda
from boda_amd.rtc import make_rtc, RtcArg, RtcFuncCall, RtcFuncInfo
from boda_amd import gen_data as gd

def conv_op(B, C, H, OC):
    d = lambda n, s: Nda(Dims(n, s, "float")); none = lambda yx: Nda(Dims(("y", "x"), tuple(yx), "none"), "none")
    return Op({"type": "Convolution"}, {"in": d(("img", "chan", "y", "x"), (B, C, H, H)), "filts": d(("out_chan", "in_chan", "y", "x"), (OC, C, 1, 1)),
               "biases": d(("out_chan",), (OC,)), "out": d(("img", "chan", "y", "x"), (B, OC, H, H)), "stride": none((1, 1)), "in_pad": none((0, 0)),
               "kern_sz": none((1, 1)), "out_chans": Nda(None, "uint32_t", (OC,))})

rtc = make_rtc(); rtc.init(); rtc.compile(gd.func_infos())
shapes = [tuple(int(x) for x in s.split(":")) for s in os.environ.get("SHAPES", "256:96:55:96,64:64:56:256,64:256:56:64,64:64:56:64").split(",")]
specs = os.environ.get("SPECS", "off,auto").split(",")
for si, (B, C, H, OC) in enumerate(shapes):
    op = conv_op(B, C, H, OC)
    anno = add_codegen_annotations(op, OpTune(hip_dtype=os.environ.get("DTYPE", ""))); fn = anno.get_func_name(); g = f"{fn}__{si}"
    rtc.compile([RtcFuncInfo(g, "", [x for x, _ in NATIVE_ARGS[fn]], anno)])
    am = {}
    for an, io in NATIVE_ARGS[fn]:
        if io == "REF": am[an] = RtcArg.ref(anno.get_dims(an)); continue
        vn = f"{an}_{si}"; rtc.create_var_with_dims(vn, anno.get_dims(an)); am[an] = RtcArg.var(vn)
        if io == "IN": rtc.run(gd.gen_call("Convolution", an, vn, anno.get_dims(an), 5, 0.0))
    call = RtcFuncCall(g, am)
    fl = 2.0 * B * H * H * OC * C; by = 4.0 * (B * H * H * (C + OC) + OC * C + OC)
    ref = None
    tiles = os.environ.get("TILES", "").split(",") if os.environ.get("TILES") else None
    for spec in (tiles or specs):
        try:
            if tiles: rtc.set_tune("conv_tile", "" if spec == "auto" else spec)
            else: rtc.set_tune("k1_stream", "" if spec == "auto" else spec)
            rtc.set_var_to_zero(f"out_{si}")
            for _ in range(int(os.environ.get('SETTLE', '400'))): rtc.run(call)
            rtc.finish_and_sync(); rtc.release_per_call_id_data()
            ids = [rtc.run(call) for _ in range(50)]; rtc.finish_and_sync()
            ms = np.array([rtc.get_dur(c, c) for c in ids]); rtc.release_per_call_id_data()
            out = rtc.copy_var_to_nda(f"out_{si}") if hasattr(rtc, "copy_var_to_nda") else None
            ll = rtc.last_launch(); li = f"{ll.get('kernel')} {ll.get('tile', ll.get('cfg'))} grid {ll.get('grid')}"
            same = "ref" if ref is None else ("SAME" if np.array_equal(ref, out) else f"DIFF({int((ref != out).sum())}, max {float(np.abs(ref - out).max()):.3g})")
            if ref is None: ref = out
            print(f"B{B} C{C} {H}x{H} OC{OC} spec={spec:12s} mean {ms.mean()*1e3:7.1f} us min {ms.min()*1e3:7.1f} us  {fl/ms.mean()/1e9:6.1f} TF/s {by/ms.mean()/1e6:6.0f} GB/s  {same}  {li}", flush=True)
        except Exception as e:
            print(f"B{B} C{C} {H}x{H} OC{OC} spec={spec}: {type(e).__name__}: {str(e)[:200]}", flush=True)
    for an, io in NATIVE_ARGS[fn]:
        if io != "REF": rtc.release_var(f"{an}_{si}")
