#!/usr/bin/env python3
"""Scan code objects for the gfx950 wide-store hazard found with k1_quad_f32.hip: a buffer/global store of more than 8 bytes whose data registers are
overwritten by a VALU instruction (or a load's return is irrelevant: those are counted) within WINDOW issue slots.  LLVM's hazard recogniser exempts
stores with an SGPR soffset; on MI355X that case corrupts data (profiles/r04_probe_wide_store_hazard.txt).   usage: store_hazard_audit.py [dir|files...]"""
import os, re, subprocess, sys
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
WINDOW = int(os.environ.get("WINDOW", "2"))
def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()
def audit(fn):
    txt = subprocess.run([OBJDUMP, "-d", fn], capture_output=True, text=True).stdout
    ins = []
    for line in txt.splitlines():
        line = line.split("//")[0].strip()
        if not line or line.endswith(":") or line.startswith(("Disassembly", fn)): continue
        ins.append(line)
    hits = []
    for i, l in enumerate(ins):
        m = re.match(r"(buffer_store_dwordx[34]|global_store_dwordx[34]|flat_store_dwordx[34])\s+(.*)", l)
        if not m: continue
        ops = [t.strip() for t in m.group(2).split(",")]
        data = regs(ops[1]) if m.group(1).startswith(("global", "flat")) else regs(ops[0])
        sgpr_soff = m.group(1).startswith("buffer") and bool(re.match(r"s\d+", ops[3].split()[0]))
        for j in range(i + 1, min(len(ins), i + 1 + WINDOW)):
            n = ins[j]
            if n.startswith("s_nop") or n.startswith("s_waitcnt"): break
            if n.startswith(("v_", "ds_read", "buffer_load", "global_load")) and not n.startswith(("v_cmp", "v_mfma")):
                dst = regs(n.split(None, 1)[1].split(",")[0].strip()) if " " in n else set()
                if n.startswith("v_") and dst & data:
                    hits.append((l, n, j - i, sgpr_soff)); break
    return hits
args = sys.argv[1:] or ["boda_amd/_kcache"]
files = []
for a in args:
    files += [os.path.join(a, f) for f in sorted(os.listdir(a)) if f.endswith(".hsaco")] if os.path.isdir(a) else [a]
tot = 0
for fn in files:
    h = audit(fn)
    if h:
        tot += len(h)
        print(f"{fn}: {len(h)} hit(s)")
        for st, nx, d, sg in h[:4]: print(f"    {st}\n      -> +{d}: {nx}   [{'SGPR soffset' if sg else 'no SGPR soffset'}]")
print(f"{len(files)} code objects, {tot} hit(s) within {WINDOW} slot(s)")
