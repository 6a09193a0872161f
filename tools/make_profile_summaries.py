#!/usr/bin/env python3
"""gpurun_out/<dir>/{stats,fetch,write,sq}_<key>/p_results.db + calib -> profiles/<tag>_*.txt + profiles/pmc_summary.json (each entry carries the
kernel-source hash it was measured with: bench.py reports roofline.traffic only while that hash is current)"""
import json, os, sqlite3, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "final")
tag = sys.argv[2] if len(sys.argv) > 2 else "r06"
out = os.path.join(ROOT, "profiles")
os.makedirs(out, exist_ok=True)
summ = os.path.join(ROOT, "tools", "rocprof_summary.py")


def q(db, sql):
    return list(sqlite3.connect(db).execute(sql))


# FETCH_SIZE calibration (known-bytes streaming read, tools/fetch_calib.py)
cal = q(os.path.join(src, "calib", "c_results.db"), "select avg(value) from counters_collection where counter_name='FETCH_SIZE' and kernel_name like 'calib_stream_read%'")[0][0]
known = 1 << 30
factor = known / (cal * 1024.0)
res = {"_fetch_size_calibration": {"known_bytes": known, "FETCH_SIZE_KB": cal, "bytes_per_reported_byte": round(factor, 4),
                                   "note": "16 B/lane coalesced stream; FETCH_SIZE under-reports by this factor on gfx950 (guide: exactly 2)"}}
hash_fn = os.path.join(src, "kernel_src_hash.txt")
khash = open(hash_fn).read().strip() if os.path.exists(hash_fn) else ""
WORK = {"sgemm-ops-full": ("--workload sgemm-ops-full", "bodahip_sgemm%f32"), "alexnet": ("--workload alexnet", "bodahip_%f32"), "nin": ("--workload nin", "bodahip_%f32"),   # (LIKE patterns: sgemm_f32 + sgemm_big_f32; conv_f32 + fc_f32 + k1_quad_f32)
        "googlenet-bf16-nhwc": ("--workload googlenet --dtype bf16 --layout nhwc", "bodahip_conv_nhwc%bf16"),      # (SQL LIKE pattern: the implicit-GEMM and the input-patch kernel)
        "resnet50-bf16-nhwc": ("--workload resnet50 --dtype bf16 --layout nhwc", "bodahip_conv_nhwc%bf16")}
for w, (cmdargs, kern) in WORK.items():
    sdb = os.path.join(src, f"stats_{w}", "p_results.db")
    if not os.path.exists(sdb):
        continue
    with open(os.path.join(out, f"{tag}_{w}_kernel_stats.txt"), "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats -- python bench.py {cmdargs} --steps 20 --warmup 5 --no-cpu-baseline\n")
        f.write(subprocess.check_output([sys.executable, summ, sdb, "--by-grid"], text=True))
    rows = {}
    for cn, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
        for g, v in q(os.path.join(src, f"{sub}_{w}", "p_results.db"), f"select grid_size, sum(value) from counters_collection where counter_name='{cn}' and kernel_name like '{kern}' group by grid_size"):
            rows.setdefault(g, {})[cn] = v
    sq = {}
    for g, cn, v, dur, nn in q(os.path.join(src, f"sq_{w}", "p_results.db"), f"select grid_size, counter_name, sum(value), sum(end-start), count(*) from counters_collection where kernel_name like '{kern}' group by grid_size, counter_name"):
        sq.setdefault(g, {})[cn] = v; sq[g]["_dur_ns"] = dur; sq[g]["_n"] = nn
    with open(os.path.join(out, f"{tag}_{w}_pmc.txt"), "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --pmc <set> -- python bench.py {cmdargs} --steps 1 --warmup 0   (separate passes: FETCH_SIZE | WRITE_SIZE | SQ set); kernel {kern}; kernel sources {khash}\n")
        f.write(f"# HBM bytes = FETCH_SIZE_KB*1024*{factor:.3f} (calibrated, see pmc_summary.json) + WRITE_SIZE_KB*1024 ; summed over launches of the same grid size\n")
        f.write("# grid(threads)  workgroups/launch  launches  fetch_MB(corrected)  write_MB  | mfma_busy%  clock_GHz  waves  wave_cycles: wait_inst% wait_any% active%\n")
        wgs = {}
        try:   # workgroups per launch and launches per grid size (round 5: how full the chip is, launch by launch -- 256 CUs)
            for g, wsz, nn in q(os.path.join(src, f"fetch_{w}", "p_results.db"), f"select grid_size, max(workgroup_size), count(*) from counters_collection where counter_name='FETCH_SIZE' and kernel_name like '{kern}' group by grid_size"):
                wgs[g] = (int(g // max(1, wsz)), int(nn))
        except Exception:
            pass
        tot_f = tot_w = 0.0
        for g in sorted(rows):
            fb = rows[g].get("FETCH_SIZE", 0) * 1024 * factor; wb = rows[g].get("WRITE_SIZE", 0) * 1024
            tot_f += fb; tot_w += wb
            s = sq.get(g, {})
            line = f"{g:12d}  {wgs.get(g, (0, 0))[0]:8d}  {wgs.get(g, (0, 0))[1]:6d}  {fb/1e6:14.1f}  {wb/1e6:10.1f}"
            if s.get("GRBM_GUI_ACTIVE"):
                cyc = s["GRBM_GUI_ACTIVE"] / 8.0
                wc = s.get("SQ_WAVE_CYCLES", 0) or 1
                long_enough = s["_dur_ns"] / max(1, s.get("_n", 1)) > 2e5   # (GRBM_GUI_ACTIVE spans the dispatch gaps of short launches: clock / busy only for >= 200 us)
                # short launches: GRBM_GUI_ACTIVE spans the dispatch gaps, so busy % is taken against the kernel's own duration at the nominal 2.4 GHz (a LOWER bound of
                # the busy share, marked ~) and no clock is derived
                busy_nom = 100 * s.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / 1024 / (s["_dur_ns"] * 2.4)
                line += (f"  | " + (f"{100*s.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/1024/cyc:9.1f}  {cyc/s['_dur_ns']:9.2f}" if long_enough else f"   ~{busy_nom:5.1f}          -") + f"  {int(s.get('SQ_WAVES',0)):6d}"
                         f"  {100*s.get('SQ_WAIT_INST_ANY',0)/wc:6.1f} {100*s.get('SQ_WAIT_ANY',0)/wc:6.1f} {100*s.get('SQ_ACTIVE_INST_ANY',0)/wc:6.1f}")
            f.write(line + "\n")
        # passes over the op list inside one profiled run: the timed step AND bench.dominant_kernel()'s untimed pass (round 5: every call once more, to name the kernel with
        # the largest summed time) -- the launch count of the rarest grid size.  The per-step figures are per ONE pass.
        passes = max(1, min((n for _, n in wgs.values()), default=1))
        f.write(f"# {passes} pass(es) over the op list in this run (timed step + bench.py's untimed dominant-kernel pass); rows above are sums over all of them\n")
        f.write(f"# total per step (one pass): fetch {tot_f/passes/1e9:.3f} GB (corrected), write {tot_w/passes/1e9:.3f} GB\n")
    res[w] = {"hbm_bytes_per_step": int((tot_f + tot_w) / passes), "fetch_bytes_corrected": int(tot_f / passes), "write_bytes": int(tot_w / passes), "kernel_src_hash": khash, "file": f"{tag}_{w}_pmc.txt",
              "passes_profiled": passes}
# whole-net runs: kernel trace (and, where collected, the SQ counters per kernel name) -- no HBM-byte entry in pmc_summary.json
NETS = {"googlenet-net-bf16-nhwc": "--workload googlenet-net --dtype bf16 --layout nhwc", "nin-net-b128": "--workload nin-net --batch 128"}
for w, cmdargs in NETS.items():
    sdb = os.path.join(src, f"stats_{w}", "p_results.db")
    if os.path.exists(sdb):
        with open(os.path.join(out, f"{tag}_{w}_kernel_stats.txt"), "w") as f:
            f.write(f"# rocprofv3 --kernel-trace --stats -- python bench.py {cmdargs} --steps 20 --warmup 5 --no-cpu-baseline\n")
            f.write(subprocess.check_output([sys.executable, summ, sdb, "--by-grid"], text=True))
    fdb, wdb = os.path.join(src, f"fetch_{w}", "p_results.db"), os.path.join(src, f"write_{w}", "p_results.db")
    if os.path.exists(fdb) and os.path.exists(wdb):   # round 5: HBM bytes of ONE forward pass, every kernel of it (convs, pool / LRN / layout passes): the config legs' roofline.traffic
        not_setup = "kernel_name not like 'gen_data%' and kernel_name not like '%xpose_filts%' and kernel_name not like '__amd_rocclr%'"
        fb = (q(fdb, f"select sum(value) from counters_collection where counter_name='FETCH_SIZE' and {not_setup}")[0][0] or 0) * 1024 * factor
        wb = (q(wdb, f"select sum(value) from counters_collection where counter_name='WRITE_SIZE' and {not_setup}")[0][0] or 0) * 1024
        passes = max(1, int(open(os.path.join(src, f"passes_{w}.txt")).read().strip())) if os.path.exists(os.path.join(src, f"passes_{w}.txt")) else 1
        res[w] = {"hbm_bytes_per_step": int((fb + wb) / passes), "fetch_bytes_corrected": int(fb / passes), "write_bytes": int(wb / passes), "kernel_src_hash": khash, "file": f"{tag}_{w}_pmc.txt",
                  "passes_profiled": passes, "scope": "every kernel of a forward pass except data generation and the one-time filter layout passes"}
    qdb = os.path.join(src, f"sq_{w}", "p_results.db")
    if os.path.exists(qdb):
        per = {}
        for kn, cn, v, dur, nn in q(qdb, "select kernel_name, counter_name, sum(value), sum(end-start), count(*) from counters_collection group by kernel_name, counter_name"):
            per.setdefault(kn, {})[cn] = v; per[kn]["_dur_ns"] = dur; per[kn]["_n"] = nn
        with open(os.path.join(out, f"{tag}_{w}_pmc.txt"), "w") as f:
            f.write(f"# rocprofv3 --kernel-trace --pmc <SQ set> -- python bench.py {cmdargs} --steps 1 --warmup 0 --settle-ms 0; kernel sources {khash}\n")
            f.write("# kernel  launches  avg_us(under the counters)  | mfma_busy% against the kernel's own duration at the nominal 2.4 GHz (a lower bound)  wave_cycles: wait_inst% wait_any% active%\n")
            for kn in sorted(per, key=lambda k: -per[k]["_dur_ns"]):
                s_ = per[kn]; wc = s_.get("SQ_WAVE_CYCLES", 0) or 1
                f.write(f"{kn[:48]:48s} {s_['_n']:5d} {s_['_dur_ns']/s_['_n']/1e3:9.1f}  | {100*s_.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/1024/(s_['_dur_ns']*2.4):6.1f}  "
                        f"{100*s_.get('SQ_WAIT_INST_ANY',0)/wc:6.1f} {100*s_.get('SQ_WAIT_ANY',0)/wc:6.1f} {100*s_.get('SQ_ACTIVE_INST_ANY',0)/wc:6.1f}\n")
import glob, shutil
for fn in glob.glob(os.path.join(src, "bench_*.json")):   # the bench lines of the same box
    if os.path.getsize(fn) > 10:
        shutil.copy(fn, os.path.join(out, f"{tag}_" + os.path.basename(fn)))
json.dump(res, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
