"""GPU tool: per-step cost breakdown of hip_conv_nhwc on chosen layers: full kernel vs ABLATE=1 (no operand loads) / 2 (no reads, no MFMAs) / 3 (no MFMAs)."""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np, bench
    from boda_amd.cnn_op import OpTune, add_codegen_annotations
    from boda_amd.ops_prof import OpsBackend, profile_rcg_call
    from boda_amd.rtc import make_rtc
    rtc = make_rtc("(be=hip)", 0); rtc.init(); be = OpsBackend(rtc)
    seen = {}
    for op in bench.net_conv_ops(os.environ.get("NET", "resnet-50"), 64): seen.setdefault(op.to_str(), op)
    ops = list(seen.values())
    out = []
    for i in [int(x) for x in os.environ.get("SEL", "3,7,8,12,14,17").split(",")]:
        t = os.environ.get("TILE", "")
        anno = add_codegen_annotations(ops[i], OpTune(hip_dtype="bf16", hip_layout="nhwc", hip_tile=t))
        _, prc = profile_rcg_call(be, anno, 5, 0.0, 12, want_outs=False, tile=t)
        out.append(f"{float(np.median(prc.all_secs[2:]))*1e6:7.1f}")
    print(os.environ.get("BODAHIP_EXTRA_DEFS", "full").ljust(14), prc.launch["cfg"], " ".join(out), flush=True)
else:
    for ab in ("", "-DABLATE=1", "-DABLATE=2", "-DABLATE=3", "-DABLATE=4", "-DABLATE=5", "-DABLATE=6"):
        env = dict(os.environ); env["BODAHIP_EXTRA_DEFS"] = ab
        if not ab: env.pop("BODAHIP_EXTRA_DEFS")
        env["BODAHIP_CACHE_DIR"] = "/tmp/kc_ablate" + ab.replace("-D", "_").replace("=", "")
        subprocess.run([sys.executable, __file__, "child"], env=env)
