"""oracle/ref_cucl.py -- TEST INFRASTRUCTURE: builds the reference's OWN convolution / sgemm kernels for gfx950 into oracle/_ref/cucl/.

The reference's arithmetic exists only as CUCL templates (test/rtc/{sgemm,conv,k1conv,tconv,xpose_filts,k1conv_xpose_in,tconv_xpose_in}.cucl
of a Boda checkout) that its code generator instantiates per op and hands to NVRTC / OpenCL.  Where a checkout is present (the build
container: /root/reference), this recipe instantiates them with this repository's restatement of that generator
(boda_amd/cucl_template.py + oracle/cnn_codegen.py), compiles each generated function with hiprtc for gfx950, and writes
    oracle/_ref/cucl/<function>.hsaco      code objects (git-ignored; they travel to the GPU box like any built .so)
    oracle/_ref/cucl/manifest.json         per op: the op line, the tune, per function its name, argument list and kinds, tpb, blks
Nothing of the reference's text is kept: the generated sources exist only in memory.  On the GPU box tests/test_gpu_ref_cucl.py loads the
code objects through be=hip (bodahip_compile_code_object), runs the reference's layout passes and kernels on the reference's
deterministic data and holds the results to the oracle -- the real reference, checked and timed on the same silicon as the native kernels.
Only tests/ and __graft_entry__.build() use this module."""
from __future__ import annotations
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(HERE, "_ref", "cucl")
REF_RTC_DIR = "/root/reference/test/rtc"


def workloads():
    """(tag, op, tune) of everything that is built: BASELINE config 2 sizes for sgemm, configs 3 / 4 layers for the conv variants, and the small
    ops whose reference digests the repository holds (tests/golden/wisdom/conv-debug.wis)."""
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import bench
    from boda_amd.cnn_op import OpTune
    from boda_amd.op import read_ops
    res = []
    for op in bench.sgemm_full_ops():
        n = op.sgemm_geom()["M"]
        if n in (256, 2048, 4096, 8192):
            res.append((f"sgemm{n}", op, OpTune()))
    for op in bench.sgemm_full_ops():      # the other sgemm variants (op_tune use_local_mem = 0 / 2 / 3; vw = 4: HIP has float4, not OpenCL's float8)
        n = op.sgemm_geom()["M"]
        if n in (256, 2048):
            for lm, name in ((0, "nolocal"), (2, "simd"), (3, "simdlocal")):
                res.append((f"sgemm{n}_{name}", op, OpTune(use_local_mem=lm, vw=4)))
    kt = OpTune(k1conv=1, tconv=1)
    for b in (2, 256):
        for i, op in enumerate(bench.alexnet_b256_ops(b)):
            res.append((f"alexnet_b{b}_l{i}", op, kt))
        for i, op in enumerate(bench.nin_ops(b)):
            if i in (1, 4, 7, 10):      # the 1x1 layers the reference runs as k1conv (cccp1 / 3 / 5 / 7)
                res.append((f"nin_b{b}_l{i}", op, kt))
    # the inner-product variant for the layers with a 1x1 output.  Its local-memory loads are unguarded in the reference ("can load garbage",
    # src/cnn_codegen.cc:226,240): only blockings that divide images and out_chans exactly are safe -- fc6 / fc7 at 64 and 256 images (fc8's 1000
    # out_chans would read 24 filters past the end of the tensor)
    st = OpTune(k1conv=1, tconv=1, use_local_mem=2, vw=4)      # k1conv_simd: vector loads, no local memory, in / filts / out as padded (chan, pel) matrices
    for b in (2, 256):
        for i in (4, 7):
            res.append((f"nin_b{b}_l{i}_simd", bench.nin_ops(b)[i], st))
    ct = OpTune(use_local_mem=2, vw=4, Kb=1)      # conv_simd: the general vector variant (stride 4 / 11x11, 5x5 pad 2, 3x3 pad 1)
    for b in (2, 64):
        for i in (0, 1, 2):
            res.append((f"alexnet_b{b}_l{i}_simd", bench.alexnet_b256_ops(b)[i], ct))
    it = OpTune(k1conv=1, tconv=1, ipconv=1)
    for b in (64, 256):
        for i in (5, 6):
            res.append((f"alexnet_b{b}_l{i}_ip", bench.alexnet_b256_ops(b)[i], it))
    for i, op in enumerate(read_ops(os.path.join(ROOT, "tests", "golden", "ops", "conv-ops-debug.txt"))):
        res.append((f"debug{i}", op, kt)); res.append((f"debug{i}_plain", op, OpTune()))
    return res


def build(rtc_dir: str = REF_RTC_DIR, verbose: bool = False) -> int:
    """-> number of code objects written (0 when no Boda checkout is present: the prebuilt files, if any, are left alone)."""
    if not os.path.isdir(rtc_dir):
        return 0
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from boda_amd import rtc
    from oracle import cnn_codegen as cc
    from boda_amd.cucl_template import load_template
    os.makedirs(OUT, exist_ok=True)
    manifest, n = [], 0

    def emit(fn_name: str, template: str, anno):
        nonlocal n
        inst = cc.instantiate_ref(rtc_dir, template, anno, fn_name)
        path = os.path.join(OUT, fn_name + ".hsaco")
        rtc.compile_to_file(inst.src, path)
        n += 1
        t = load_template(rtc_dir, template)
        kinds = {ad.vn: ("SCALAR" if ad.loi == 0 else ad.io_type) for ad in t.arg_decls}
        multi = [ad.vn for ad in t.arg_decls if ad.multi]
        def kind(a):      # (the members <vn>_<i> of a _multi pack carry their declaration's kind)
            return kinds[a] if a in kinds else next(kinds[m] for m in multi if a.startswith(m + "_") and a[len(m) + 1:].isdigit())
        return {"func": fn_name, "template": template, "arg_names": inst.arg_names, "arg_kinds": [kind(a) for a in inst.arg_names], "tpb": inst.tpb,
                "blks": inst.blks, "file": os.path.basename(path)}

    for tag, op, tune in workloads():
        anno = cc.annotate_ref(op, tune)
        entry = {"tag": tag, "op": op.to_str(), "tune": tune.to_str(), "variant": anno.get_func_name(), "xposes": []}
        for tname, src_arg, dst_arg, xop in cc.xpose_ops(anno):
            e = emit(f"{tag}__{tname}", tname, xop); e.update(src=src_arg, dst=dst_arg); entry["xposes"].append(e)
        entry["main"] = emit(f"{tag}__{anno.get_func_name()}", anno.get_func_name(), anno)
        entry["post"] = []
        for tname, src_arg, dst_arg, xop in cc.post_xpose_ops(anno):      # (variants that write a transposed out: <func>_xpose_out afterwards)
            e = emit(f"{tag}__{tname}", tname, xop); e.update(src=src_arg, dst=dst_arg); entry["post"].append(e)
        manifest.append(entry)
        if verbose:
            print(tag, entry["variant"], entry["main"]["tpb"], entry["main"]["blks"])
    # write-xposed chaining (src/rtc_fwd.cc:495-503): NiN cccp1 -> cccp2 as two k1conv functions, the first writing the second's input layout (no
    # k1conv_xpose_in between them)
    import bench
    from boda_amd.cnn_op import OpTune
    for b in (2, 256):
        ops2 = bench.nin_ops(b)[1:3]; kt = OpTune(k1conv=1, tconv=1)
        a1, a2 = cc.annotate_ref(ops2[0], kt), cc.annotate_ref(ops2[1], kt)
        cc.chain_k1conv(a1, a2)
        tag = f"nin_b{b}_chain_l1l2"
        l1 = {"xposes": [dict(emit(f"{tag}__1_{t}", t, x), src=sa, dst=da) for t, sa, da, x in cc.xpose_ops(a1)], "main": emit(f"{tag}__1_k1conv_wx", "k1conv", a1)}
        l2 = {"xposes": [dict(emit(f"{tag}__2_{t}", t, x), src=sa, dst=da) for t, sa, da, x in cc.xpose_ops(a2) if da != "in"], "main": emit(f"{tag}__2_k1conv", "k1conv", a2)}
        manifest.append({"tag": tag, "variant": "k1conv_chain", "ops": [o.to_str() for o in ops2], "tune": kt.to_str(), "l1": l1, "l2": l2})
    # the one template of the reference with a `_multi` argument pack (test/rtc/reduce.cucl: out = sum of ins_num tensors; custom code generation
    # src/cnn_codegen.cc:28-34): exercises the pack expansion of the template layer on the GPU
    from boda_amd.op import Dims, Nda, Op
    d = Dims.make("float", img=3, chan=5, y=7, x=9)
    rvals = {"out": Nda(d), "ins_num": Nda(Dims((), (), "uint32_t"), "uint32_t", (3,))}
    rvals.update({f"ins_{i}": Nda(d) for i in range(3)})
    rop = Op({"type": "Reduce", "func_name": "reduce"}, rvals)
    manifest.append({"tag": "reduce3", "op": rop.to_str(), "tune": "()", "variant": "reduce", "xposes": [], "main": emit("reduce3__reduce", "reduce", rop)})
    with open(os.path.join(OUT, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1)
    return n


if __name__ == "__main__":
    print(build(verbose=True), "code objects in", OUT)
