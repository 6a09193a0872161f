#!/usr/bin/env python3
"""Where does the quad 1x1 kernel differ from the oracle?  CASE=B:C:H:W:OC SPEC=q4x3x8"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from boda_amd.op import Op, Dims
from boda_amd.cnn_op import OpTune, add_codegen_annotations
from boda_amd.rtc import make_rtc, RtcArg, RtcFuncCall, RtcFuncInfo
from boda_amd.ops_prof import NATIVE_ARGS
from oracle import boda_oracle as bo
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_parity import _conv_op
B, C, H, W, OC = [int(x) for x in os.environ.get("CASE", "3:96:11:9:96").split(":")]
spec = os.environ.get("SPEC", "q4x3x8")
rtc = make_rtc(); rtc.init()
op = _conv_op(B, C, H, W, OC, 1, 1, 1, 0)
anno = add_codegen_annotations(op, OpTune()); fn = anno.get_func_name()
rtc.compile([RtcFuncInfo("k", "", [a for a, _ in NATIVE_ARGS[fn]], anno)])
x = bo.gen_conv_in(B, C, H, W); f = bo.gen_conv_filts(OC, C, 1, 1); b = bo.gen_conv_biases(OC)
names = {"in": ("k_in", anno.get_dims("in"), x), "filts": ("k_f", anno.get_dims("filts"), f), "biases": ("k_b", anno.get_dims("biases"), b),
         "out": ("k_out", anno.get_dims("out"), np.full(anno.get_dims("out").sizes, 7.0, np.float32))}
for vn, d, arr in names.values():
    rtc.create_var_with_dims(vn, d); rtc.copy_nda_to_var(vn, arr)
am = {an: RtcArg.var(names[an][0]) for an in names}
am["stride"] = RtcArg.ref(anno.get_dims("stride")); am["in_pad"] = RtcArg.ref(anno.get_dims("in_pad")); am["out_chan_off"] = RtcArg.scalar(0, "uint32_t")
rtc.set_tune("k1_stream", spec)
rtc.run(RtcFuncCall("k", am)); rtc.finish_and_sync()
print(rtc.last_launch())
got = rtc.copy_var_to_nda("k_out").reshape(B, OC, H * W)
want = bo.conv_fwd(x, f, b, (1, 1), (0, 0), True).reshape(B, OC, H * W)
bad = np.argwhere(got != want)
print("bad", len(bad), "of", got.size)
if len(bad):
    print("imgs", np.unique(bad[:, 0])); print("chans", np.unique(bad[:, 1])); print("pels", np.unique(bad[:, 2]))
    raw = bo.conv_fwd(x, f, b, (1, 1), (0, 0), False).reshape(B, OC, H * W)
    nob = bo.conv_fwd(x, f, np.zeros_like(b), (1, 1), (0, 0), False).reshape(B, OC, H * W)
    for i, o, p in bad[:10]:
        m = np.argwhere(raw[i] == got[i, o, p]); m2 = np.argwhere(nob[i] == got[i, o, p])
        print(i, o, p, got[i, o, p], want[i, o, p], "raw-match (chan,pel):", m[:3].tolist(), "no-bias match:", m2[:3].tolist())
