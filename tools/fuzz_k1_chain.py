#!/usr/bin/env python3
"""Random-shape parity sweep of hip_conv_k1_chain (two chained 1x1 convolutions, kernels/k1_quad_f32.hip -DCHAIN=1) against the CPU oracle's two layers, bit-exact:
in_chans 1..128, intermediate channels 1..96, out_chans 1..128, planes of 4 pels and up, every ReLU combination, with / without the intermediate tensor, into a channel
slice of a wider output with a guard band.   usage: fuzz_k1_chain.py [n_cases] [seed]    (GPU box; exit code 1 on any mismatch)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from boda_amd.cnn_op import NATIVE_ARGS, OpTune, add_codegen_annotations, annotate_k1_chain, k1_chain_applies
from boda_amd.op import Dims, parse_op
from boda_amd.rtc import make_rtc, RtcArg, RtcFuncCall, RtcFuncInfo
from oracle import boda_oracle as bo

def conv_op(B, C, H, W, OC):
    return parse_op(f"(str_vals=(type=Convolution),nda_vals=(biases=(dims=(out_chan={OC})),filts=(dims=(out_chan={OC},in_chan={C},y=1,x=1)),in=(dims=(img={B},chan={C},y={H},x={W})),"
                    f"in_pad=(tn=none,dims=(y=0,x=0)),kern_sz=(tn=none,dims=(y=1,x=1)),out=(dims=(img={B},chan={OC},y={H},x={W})),out_chans=(tn=uint32_t,v={OC}),stride=(tn=none,dims=(y=1,x=1))))")

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
rtc = make_rtc(); rtc.init()
bad = 0
for it in range(n):
    B = int(rng.choice([1, 2, 3, 5, 9])); C = int(rng.choice([1, 2, 3, 5, 8, 16, 33, 64, 96, 127, 128])); MID = int(rng.integers(1, 97)); OC2 = int(rng.integers(1, 129))
    H = int(rng.integers(1, 60)); W = int(rng.integers(1, 60))
    if H * W < 4: W = 4
    if 2.0 * B * H * W * (MID * C + OC2 * MID) > 2e9: continue
    relus = (int(rng.integers(0, 2)), int(rng.integers(0, 2))); with_mid = bool(rng.integers(0, 2)); off = int(rng.integers(0, 4))
    a = add_codegen_annotations(conv_op(B, C, H, W, MID), OpTune()); b = add_codegen_annotations(conv_op(B, MID, H, W, OC2), OpTune())
    if not k1_chain_applies(a, b): continue
    ch = annotate_k1_chain(a, b, *relus)
    args = [x for x, _ in NATIVE_ARGS["hip_conv_k1_chain"]] + (["mid"] if with_mid else [])
    rtc.compile([RtcFuncInfo("fz", "", args, ch)])
    x = bo.gen_conv_in(B, C, H, W); f1 = bo.gen_conv_filts(MID, C, 1, 1); b1 = bo.gen_conv_biases(MID)
    f2 = (bo.gen_conv_filts(OC2, MID, 1, 1) * np.float32(0.25)).astype(np.float32); b2 = bo.gen_conv_biases(OC2)
    wide = Dims.make("float", img=B, chan=OC2 + off + 3, y=H, x=W)
    names = {"in": ("fz_in", a.get_dims("in"), x), "filts": ("fz_f1", a.get_dims("filts"), f1), "biases": ("fz_b1", a.get_dims("biases"), b1), "filts2": ("fz_f2", b.get_dims("filts"), f2),
             "biases2": ("fz_b2", b.get_dims("biases"), b2), "out": ("fz_out", wide, np.full(wide.sizes, 7.0, np.float32)), "mid": ("fz_mid", a.get_dims("out"), np.full(a.get_dims("out").sizes, 3.0, np.float32))}
    for vn, d, arr in names.values():
        rtc.create_var_with_dims(vn, d); rtc.copy_nda_to_var(vn, arr)
    am = {an: RtcArg.var(names[an][0]) for an in names if an != "mid" or with_mid}
    am["stride"] = RtcArg.ref(a.get_dims("stride")); am["in_pad"] = RtcArg.ref(a.get_dims("in_pad")); am["out_chan_off"] = RtcArg.scalar(off, "uint32_t")
    rtc.run(RtcFuncCall("fz", am)); rtc.finish_and_sync()
    got = rtc.copy_var_to_nda("fz_out"); mid_g = rtc.copy_var_to_nda("fz_mid")
    mid_w = bo.conv_fwd(x, f1, b1, (1, 1), (0, 0), bool(relus[0])); want = bo.conv_fwd(mid_w, f2, b2, (1, 1), (0, 0), bool(relus[1]))
    ok = np.array_equal(got[:, off:off + OC2], want) and (got[:, :off] == 7).all() and (got[:, off + OC2:] == 7).all() and (np.array_equal(mid_g, mid_w) if with_mid else (mid_g == 3).all())
    if not ok:
        bad += 1; print(f"MISMATCH B{B} C{C} {H}x{W} MID{MID} OC{OC2} relus{relus} mid{with_mid} off{off}: {int((got[:, off:off + OC2] != want).sum())} outputs differ", flush=True)
    for vn, _, _ in names.values(): rtc.release_var(vn)
    rtc.release_func("fz"); rtc.release_per_call_id_data()
print(f"hip_conv_k1_chain: {n} draws, {bad} mismatches")
sys.exit(1 if bad else 0)
