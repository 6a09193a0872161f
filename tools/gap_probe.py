import os, sys, time
if os.environ.get("WITH_TORCH"):
    import torch; torch.cuda.set_device(0); torch.cuda.synchronize()
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from boda_amd.conv_pipe import ConvPipeFwd, nin_imagenet
from boda_amd import gen_data as gd
from boda_amd.rtc import make_rtc
rtc = make_rtc(); rtc.init()
cp = nin_imagenet(256); fwd = ConvPipeFwd(rtc); fwd.init(cp)
rtc.run(gd.gen_call("Convolution", "in", "data", cp.nodes["data"], 5, 0.0)); rtc.finish_and_sync(); rtc.release_per_call_id_data()
for it in range(3):
    t=time.perf_counter()
    ids = [rtc.run(c.rfc) for c in fwd.fwd_calls]
    t1=time.perf_counter()
    rtc.finish_and_sync()
    t2=time.perf_counter()
    print(f"iter {it}: enqueue {1e3*(t1-t):.2f} ms, total wall {1e3*(t2-t):.2f} ms, first-to-last {rtc.get_dur(ids[0], ids[-1]):.3f} ms")
    if it == 2:
        prev_end = 0.0
        for c, i in zip(fwd.fwd_calls, ids):
            end = rtc.get_dur(ids[0], i); dur = rtc.get_dur(i, i)
            print(f"  {c.tag:8s} start {end-dur:8.3f} dur {dur:7.3f} gap_before {end-dur-prev_end:7.3f}")
            prev_end = end
    rtc.release_per_call_id_data()
