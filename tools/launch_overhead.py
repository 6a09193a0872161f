#!/usr/bin/env python3
"""Host-side cost of one rtc.run() (enqueue only) for a native conv, a native sgemm and a generic CUCL kernel."""
import os, sys, time
if os.environ.get("WITH_TORCH"):
    import torch; torch.cuda.set_device(0); torch.cuda.synchronize()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from boda_amd.cnn_op import OpTune, add_codegen_annotations, NATIVE_ARGS
from boda_amd.op import parse_op, Dims, Op
from boda_amd.rtc import make_rtc, RtcArg, RtcFuncCall, RtcFuncInfo
from boda_amd import gen_data as gd
rtc = make_rtc(); rtc.init(); rtc.compile(gd.func_infos())
op = parse_op("(str_vals=(type=Convolution),nda_vals=(biases=(dims=(out_chan=32)),filts=(dims=(out_chan=32,in_chan=8,y=3,x=3)),in=(dims=(img=1,chan=8,y=8,x=8)),in_pad=(tn=none,dims=(y=1,x=1)),kern_sz=(tn=none,dims=(y=3,x=3)),out=(dims=(img=1,chan=32,y=8,x=8)),out_chans=(tn=uint32_t,v=32),stride=(tn=none,dims=(y=1,x=1))))")
anno = add_codegen_annotations(op, OpTune()); fn = anno.get_func_name()
rtc.compile([RtcFuncInfo("c0", "", [a for a, _ in NATIVE_ARGS[fn]], anno)])
am = {}
for an, io in NATIVE_ARGS[fn]:
    if io == "REF": am[an] = RtcArg.ref(anno.get_dims(an))
    else: rtc.create_var_with_dims(an, anno.get_dims(an)); am[an] = RtcArg.var(an)
rfc = RtcFuncCall("c0", am)
g = gd.gen_call("Convolution", "biases", "biases", anno.get_dims("biases"), 5, 0.0)
for name, call in (("native conv", rfc), ("generic CUCL (gen_data biases)", g)):
    for _ in range(20): rtc.run(call)
    rtc.finish_and_sync(); rtc.release_per_call_id_data()
    N = 2000
    t = time.perf_counter()
    for _ in range(N): rtc.run(call)
    t1 = time.perf_counter(); rtc.finish_and_sync(); t2 = time.perf_counter()
    print(f"{name}: enqueue {1e6*(t1-t)/N:.1f} us/call, drained after +{1e3*(t2-t1):.2f} ms")
    rtc.release_per_call_id_data()
