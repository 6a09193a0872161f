#!/usr/bin/env python3
"""Scan code objects for SERIAL memory round trips: backward branches whose loop body holds a VMEM load and an s_waitcnt vmcnt(0) (every trip of the loop
costs a full memory latency -- the filter staging loops of the streaming 1x1 kernels before round 4c were 36-72 such trips at the head of every launch), and
straight-line runs of load -> vmcnt(0) -> load -> vmcnt(0) ...   usage: serial_load_audit.py [dir|files...]   (MINCHAIN=3: shortest straight-line run reported)"""
import os, re, subprocess, sys
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
MINCHAIN = int(os.environ.get("MINCHAIN", "3"))
LOAD = re.compile(r"^(buffer_load|global_load|flat_load|scratch_load)")
def audit(fn):
    txt = subprocess.run([OBJDUMP, "-d", fn], capture_output=True, text=True).stdout
    kern, out, ins = "?", [], []
    def flush():
        if not ins: return
        addr = {a: i for i, (a, _) in enumerate(ins)}
        # loops: a branch to a lower address
        for i, (a, l) in enumerate(ins):
            m = re.match(r"s_cbranch_\w+\s+\S+\s*(?:;|//)?.*?<[^>]*\+0x([0-9a-f]+)>", l) or re.match(r"s_branch\s+\S+.*?<[^>]*\+0x([0-9a-f]+)>", l)
            if not m: continue
        # straight-line chains
        run, last_wait = 0, False
        seen_load = False
        for a, l in ins:
            if LOAD.match(l) and "lds" not in l: seen_load = True
            elif l.startswith("s_waitcnt") and "vmcnt(0)" in l:
                if seen_load: run += 1
                seen_load = False
            elif l.startswith(("s_cbranch", "s_branch", "s_barrier", "s_endpgm", "v_mfma")):
                if run >= MINCHAIN: out.append((kern, "chain", run))
                run = 0; seen_load = False
        if run >= MINCHAIN: out.append((kern, "chain", run))
    lines = txt.splitlines()
    # loops by label positions: objdump prints "<name>:" labels only for symbols; use branch target offsets instead
    cur = []
    for line in lines:
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
        if m:
            flush(); ins.clear(); kern = m.group(1); continue
        m = re.match(r"^\s+(\S.*?)\s*//\s*([0-9A-Fa-f]+):", line)
        if m: ins.append((int(m.group(2), 16), m.group(1).strip()))
    flush()
    # loops: need target address: objdump -d prints "s_cbranch_execnz 65500" style (simm16) -> compute target = addr + 4 + simm16*4
    ins2 = []
    kern = "?"
    res = list(out)
    for line in lines:
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
        if m: kern = m.group(1); ins2.append(("L", kern, 0)); continue
        m = re.match(r"^\s+(\S.*?)\s*//\s*([0-9A-Fa-f]+):", line)
        if m: ins2.append(("I", m.group(1).strip(), int(m.group(2), 16)))
    kern = "?"; body = []
    for i, (t, l, a) in enumerate(ins2):
        if t == "L": kern = l; continue
        m = re.match(r"s_c?branch\w*\s+(\d+)", l)
        if not m: continue
        simm = int(m.group(1)); simm -= 65536 if simm >= 32768 else 0
        tgt = a + 4 + simm * 4
        if tgt > a: continue
        seg = [x for x in ins2 if x[0] == "I" and tgt <= x[2] <= a]
        nload = sum(1 for x in seg if LOAD.match(x[1]) and "lds" not in x[1]); nwait0 = sum(1 for x in seg if x[1].startswith("s_waitcnt") and "vmcnt(0)" in x[1])
        nmfma = sum(1 for x in seg if x[1].startswith("v_mfma"))
        if nload and nwait0 and not nmfma: res.append((kern, f"loop of {len(seg)} instrs, {nload} load(s), {nwait0} vmcnt(0) wait(s), no MFMA", 0))
    return res
args = sys.argv[1:] or ["boda_amd/_kcache"]
files = []
for a in args:
    files += [os.path.join(a, f) for f in sorted(os.listdir(a)) if f.endswith(".hsaco")] if os.path.isdir(a) else [a]
agg = {}
for fn in files:
    for k, what, n in audit(fn):
        key = (re.sub(r"__\d+$", "", k), what if n == 0 else f"straight-line chain of {n} load->vmcnt(0) round trips")
        agg.setdefault(key, []).append(os.path.basename(fn))
for (k, what), fs in sorted(agg.items()):
    print(f"{k}: {what}   [{len(fs)} code object(s), e.g. {fs[0]}]")
print(f"{len(files)} code objects scanned")
