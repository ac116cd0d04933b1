#!/usr/bin/env python3
"""Per-basic-block instruction mix of one kernel in a hipcc -S listing (no GPU needed).
   hipcc -O3 --offload-arch=gfx950 -std=c++17 -S --cuda-device-only x.hip -o x.s && tools/isa_stats.py x.s <mangled-name-substring>"""
import re, sys
src, pat = sys.argv[1], sys.argv[2]
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^[_A-Za-z0-9]+:", l) and pat in l)
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
blocks, cur = [], ["entry", []]
for l in lines[start + 1:end + 1]:
    s = l.strip()
    if not s or s.startswith(";") or s.startswith("."):
        if re.match(r"^\.LBB[0-9_]+:", s):
            blocks.append(cur); cur = [s.rstrip(":"), []]
        continue
    cur[1].append(s.split()[0])
blocks.append(cur)
def cls(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("ds_"): return "lds"
    if op.startswith("global_load_lds") or "lds" in op and op.startswith("buffer_load"): return "dma"
    if op.startswith(("global_load", "buffer_load")): return "vld"
    if op.startswith(("global_store", "buffer_store")): return "vst"
    if op.startswith("scratch_"): return "scratch"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_"): return "salu"
    if op.startswith("v_"): return "valu"
    return "other"
tot = {}
for name, ops in blocks:
    c = {}
    for o in ops:
        c[cls(o)] = c.get(cls(o), 0) + 1
        tot[cls(o)] = tot.get(cls(o), 0) + 1
    if len(ops) >= int(sys.argv[3]) if len(sys.argv) > 3 else 20:
        top = {}
        for o in ops:
            if cls(o) == "valu": top[o] = top.get(o, 0) + 1
        tv = sorted(top.items(), key=lambda kv: -kv[1])[:8]
        print(f"{name:12s} n={len(ops):5d} " + " ".join(f"{k}={v}" for k, v in sorted(c.items())) + "\n             " + " ".join(f"{k}:{v}" for k, v in tv))
print("TOTAL", tot)
