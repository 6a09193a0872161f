#!/usr/bin/env python3
"""hip_conv_k1_chain (k1_quad_f32.hip -DCHAIN=1: two 1x1 convolutions, the intermediate tensor in registers) against the two hip_conv launches: per-launch time and
   bit-equality of the final (and, with MIDOUT=1, the intermediate) tensor.   SHAPES="B:C:H:MID:OC2,..." """
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from boda_amd.cnn_op import NATIVE_ARGS, OpTune, add_codegen_annotations, annotate_k1_chain
from boda_amd.op import Op, Dims, Nda
from boda_amd.rtc import make_rtc, RtcArg, RtcFuncCall, RtcFuncInfo
from boda_amd import gen_data as gd

def conv_op(B, C, H, OC):
    d = lambda n, s: Nda(Dims(n, s, "float")); none = lambda yx: Nda(Dims(("y", "x"), tuple(yx), "none"), "none")
    return Op({"type": "Convolution"}, {"in": d(("img", "chan", "y", "x"), (B, C, H, H)), "filts": d(("out_chan", "in_chan", "y", "x"), (OC, C, 1, 1)),
               "biases": d(("out_chan",), (OC,)), "out": d(("img", "chan", "y", "x"), (B, OC, H, H)), "stride": none((1, 1)), "in_pad": none((0, 0)),
               "kern_sz": none((1, 1)), "out_chans": Nda(None, "uint32_t", (OC,))})

rtc = make_rtc(); rtc.init(); rtc.compile(gd.func_infos())
shapes = [tuple(int(x) for x in s.split(":")) for s in os.environ.get("SHAPES", "128:96:55:96:96,256:96:55:96:96").split(",")]
midout = int(os.environ.get("MIDOUT", "0"))
for si, (B, C, H, MID, OC2) in enumerate(shapes):
    a = add_codegen_annotations(conv_op(B, C, H, MID), OpTune()); b = add_codegen_annotations(conv_op(B, MID, H, OC2), OpTune())
    ch = annotate_k1_chain(a, b, 1, 1)
    fa, fb, fc = f"hip_conv__a{si}", f"hip_conv__b{si}", f"hip_conv_k1_chain__{si}"
    rtc.compile([RtcFuncInfo(fa, "", [x for x, _ in NATIVE_ARGS["hip_conv"]], a), RtcFuncInfo(fb, "", [x for x, _ in NATIVE_ARGS["hip_conv"]], b),
                 RtcFuncInfo(fc, "", [x for x, _ in NATIVE_ARGS["hip_conv_k1_chain"]] + (["mid"] if midout else []), ch)])
    v = lambda n: f"{n}_{si}"
    for n, d in (("in", a.get_dims("in")), ("f1", a.get_dims("filts")), ("b1", a.get_dims("biases")), ("mid", a.get_dims("out")), ("f2", b.get_dims("filts")), ("b2", b.get_dims("biases")),
                 ("out", b.get_dims("out")), ("mid_c", a.get_dims("out")), ("out_c", b.get_dims("out"))):
        rtc.create_var_with_dims(v(n), d)
    for n, arg, d in (("in", "in", a.get_dims("in")), ("f1", "filts", a.get_dims("filts")), ("b1", "biases", a.get_dims("biases")), ("f2", "filts", b.get_dims("filts")), ("b2", "biases", b.get_dims("biases"))):
        rtc.run(gd.gen_call("Convolution", arg, v(n), d, 5, 0.0))
    R = RtcArg
    ca = RtcFuncCall(fa, {"filts": R.var(v("f1")), "biases": R.var(v("b1")), "in": R.var(v("in")), "stride": R.ref(a.get_dims("stride")), "in_pad": R.ref(a.get_dims("in_pad")), "out": R.var(v("mid"))})
    cb = RtcFuncCall(fb, {"filts": R.var(v("f2")), "biases": R.var(v("b2")), "in": R.var(v("mid")), "stride": R.ref(b.get_dims("stride")), "in_pad": R.ref(b.get_dims("in_pad")), "out": R.var(v("out"))})
    am = {"filts": R.var(v("f1")), "biases": R.var(v("b1")), "filts2": R.var(v("f2")), "biases2": R.var(v("b2")), "in": R.var(v("in")), "stride": R.ref(a.get_dims("stride")),
          "in_pad": R.ref(a.get_dims("in_pad")), "out": R.var(v("out_c"))}
    if midout: am["mid"] = R.var(v("mid_c"))
    cc = RtcFuncCall(fc, am)
    def timeit(calls, n=50, settle=int(os.environ.get("SETTLE", "300"))):
        for _ in range(settle):
            for c in calls: rtc.run(c)
        rtc.finish_and_sync(); rtc.release_per_call_id_data()
        ids = [[rtc.run(c) for c in calls] for _ in range(n)]; rtc.finish_and_sync()
        ms = np.array([sum(rtc.get_dur(i, i) for i in row) for row in ids]); rtc.release_per_call_id_data()
        return ms
    t2 = timeit([ca, cb]); ll2 = rtc.last_launch()
    t1 = timeit([cc]); ll1 = rtc.last_launch()
    out, out_c = rtc.copy_var_to_nda(v("out")), rtc.copy_var_to_nda(v("out_c"))
    same = "SAME" if np.array_equal(out, out_c) else f"DIFF({int((out != out_c).sum())} of {out.size}, max {float(np.abs(out - out_c).max()):.3g})"
    if midout:
        mid, mid_c = rtc.copy_var_to_nda(v("mid")), rtc.copy_var_to_nda(v("mid_c"))
        same += " mid:" + ("SAME" if np.array_equal(mid, mid_c) else f"DIFF({int((mid != mid_c).sum())})")
    fl = 2.0 * B * H * H * (MID * C + OC2 * MID)
    print(f"B{B} C{C} {H}x{H} MID{MID} OC{OC2}: two launches {t2.mean()*1e3:7.1f} us (min {t2.min()*1e3:.1f}) {fl/t2.mean()/1e9:6.1f} TF/s [{ll2.get('kernel')}] | chain {t1.mean()*1e3:7.1f} us (min {t1.min()*1e3:.1f}) "
          f"{fl/t1.mean()/1e9:6.1f} TF/s [{ll1.get('kernel')} grid {ll1.get('grid')}]  out:{same}", flush=True)
    for n in ("in", "f1", "b1", "mid", "f2", "b2", "out", "mid_c", "out_c"): rtc.release_var(v(n))
