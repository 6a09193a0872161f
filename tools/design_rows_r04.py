#!/usr/bin/env python3
"""Write the round-4 measurement block of DESIGN.md section 5 (between the r04-rows markers) from the committed bench lines profiles/r04_bench_*.json, so the table and
the evidence cannot drift apart.  usage: python tools/design_rows_r04.py   (after tools/make_profile_summaries.py <dir> r04)"""
import json, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def L(n):
    return json.loads(open(os.path.join(ROOT, "profiles", f"r04_bench_{n}.json")).read().strip().splitlines()[-1])
d = L("default"); co = d["conv_ops"]; tm = co["tolerance_mode"]; cf = d["configs"]
tr = d["roofline"].get("traffic")
po = lambda v: " / ".join(f"{o['tflops']:.0f}" for o in v["per_op"])
rows = ["| workload | value | roofline frac | notes |", "|---|---|---|---|"]
rows.append(f"| sgemm-ops-full fp32 (headline of the default line) | {d['value']:.1f} TF/s | **{d['roofline']['frac']:.3f}** | sizes ≥ 4096 on `sgemm_big_f32.hip`: " + " / ".join(f"{o['tflops']:.1f}" for o in d["per_op"][10:]) +
            (f"; HBM traffic {tr/1e9:.1f} GB/step (PMC)" if tr else "") + " |")
rows.append(f"| AlexNet / NiN conv-ops @256, bit-exact (`conv_ops` of the default line) | {co['alexnet']['value']:.1f} / {co['nin']['value']:.1f} TF/s | **{co['alexnet']['roofline']['frac']:.3f}** / **{co['nin']['roofline']['frac']:.3f}** | AlexNet per layer {po(co['alexnet'])} (fc6–fc8 on `fc_f32.hip`); NiN cccp1 + cccp2 on `k1_quad_f32.hip`: {co['nin']['roofline']['hbm_frac_1x1']['achieved']/1e3:.2f} TB/s = {co['nin']['roofline']['hbm_frac_1x1']['frac']:.3f} of 8 TB/s |")
rows.append(f"| … tolerance mode (`hip_exact=0`) | {tm['alexnet']['value']:.1f} / {tm['nin']['value']:.1f} | {tm['alexnet']['frac']:.3f} / {tm['nin']['frac']:.3f} | Winograd on the 3×3 layers, K slices on the fc layers |")
a, b = cf["config4_nin-net_b128_f32"], cf["config4_nin-net_b128_f32_tolerance"]
rows.append(f"| NiN full net @128 (config 4's per-GPU shard), bit-exact / tolerance | {a['value']:.1f} / {b['value']:.1f} TF/s = {a['images_per_s']/1e3:.1f} / {b['images_per_s']/1e3:.1f} k img/s | {a['roofline']['frac']:.3f} / {b['roofline']['frac']:.3f} | |")
for net, key in (("GoogLeNet", "googlenet"), ("ResNet-50", "resnet50")):
    c, i = cf[f"config5_{key}_b64_bf16_nhwc"], cf[f"config5_{key}_b64_bf16_nhwc_independent"]
    rows.append(f"| {net} list bf16 channels-last @64 (config 5 per GPU): chained graph / edge-free graph | {c['value']:.0f} / **{i['value']:.0f} TF/s** ({c['ms_per_step']:.3f} / {i['ms_per_step']:.3f} ms) | kernel time {c['roofline']['frac']:.3f}; wall {c['roofline']['timed_region']['frac']:.3f} / **{i['roofline']['timed_region']['frac']:.3f}** | bf16 parity unpinned (§4) |")
m = cf.get("config5_googlenet_b64_bf16_nhwc_independent_multi")
if m: rows.append(f"| … GoogLeNet list, edge-free graph + its implicit-GEMM members as one multi-problem launch | **{m['value']:.0f} TF/s** ({m['ms_per_step']:.3f} ms) | wall **{m['roofline']['timed_region']['frac']:.3f}** | `hip_conv_nhwc_multi` |")
g = cf["config5_googlenet-net_b64_bf16_nhwc"]
rows.append(f"| GoogLeNet full net bf16 channels-last @64 (level sets, fused poolings) | {g['images_per_s']/1e3:.1f} k img/s ({g['ms_per_step']:.3f} ms) | {g['roofline']['frac']:.3f} on the conv calls | `non_conv_ms` {g['roofline']['non_conv_ms']:.3f} |")
cb = d["cpu_baseline"]
rows.append(f"| cpu_baseline (be=cpu, {cb['cores']} threads) | {cb['value']:.2f} TF/s | — | {cb['sample']} |")
block = "<!-- r04-rows-begin -->\n**Round 4, driver command** (`python bench.py --steps 20 --warmup 5`, `profiles/r04_bench_default.json`, one box; kernel traces `profiles/r04_*_kernel_stats.txt`, PMC passes `profiles/r04_*_pmc.txt`):\n\n" + "\n".join(rows) + "\n<!-- r04-rows-end -->\n"
p = os.path.join(ROOT, "DESIGN.md"); s = open(p).read()
if "<!-- r04-rows-begin -->" in s:
    i = s.index("<!-- r04-rows-begin -->"); j = s.index("<!-- r04-rows-end -->") + len("<!-- r04-rows-end -->\n"); s = s[:i] + block + s[j:]
else:
    k = s.index("**Round 3, default line**"); s = s[:k] + block + "\n" + s[k:]
open(p, "w").write(s)
print(block)
