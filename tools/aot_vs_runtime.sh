#!/bin/bash
# GPU tool: are the code objects __graft_entry__.build() put into boda_amd/_kcache (compiled without a GPU) the ones this box would compile?  One AlexNet list run into an
# empty cache directory, every code object compared with the file of the same name in the AOT cache.
cd "$(dirname "$0")/.."; mkdir -p gpurun_out/aot; rm -rf /tmp/kc_cmp
BODAHIP_CACHE_DIR=/tmp/kc_cmp python bench.py --workload alexnet --batch 256 --steps 2 --warmup 1 --settle-ms 0 --no-cpu-baseline > /dev/null 2>&1
python - <<'P' 2>&1 | tee gpurun_out/aot/log.txt
import glob, os, hashlib, subprocess
for f in sorted(glob.glob("/tmp/kc_cmp/*.hsaco")):
    aot = os.path.join("boda_amd/_kcache", os.path.basename(f))
    a = open(f, "rb").read(); b = open(aot, "rb").read() if os.path.exists(aot) else None
    note = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f], capture_output=True, text=True).stdout
    name = [l.split()[-1] for l in note.splitlines() if ".name:" in l and "kd" not in l][:1]
    vg = [l.split()[-1] for l in note.splitlines() if ".vgpr_count" in l][:1]
    if b is None: print(os.path.basename(f), len(a), name, "vgpr", vg, "NOT IN THE AOT CACHE")
    else:
        vb = [l.split()[-1] for l in subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", aot], capture_output=True, text=True).stdout.splitlines() if ".vgpr_count" in l][:1]
        print(os.path.basename(f), len(a), name, "vgpr", vg, "same" if a == b else f"DIFFERENT (aot {len(b)} bytes, vgpr {vb})")
P
