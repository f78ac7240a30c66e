"""Static instruction statistics of one kernel in a hipcc -S listing: python scripts/isa_stats.py <file.s> <substring of mangled name>"""
import re, sys, collections
lines = open(sys.argv[1]).read().splitlines()
pat = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and pat in l.split(":")[0])
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith(".Lfunc_end"))
ins = []
for l in lines[start:end]:
    if not l.startswith("\t"): continue
    t = l.strip()
    if t.startswith((".", ";")): continue
    ins.append(t.split()[0])
c = collections.Counter(ins)
g = lambda p: sum(v for k, v in c.items() if re.match(p, k))
print(lines[start].split(":")[0][:70], "instructions", len(ins))
print("  valu", g(r"v_"), "salu", g(r"s_"), "scratch_load", g("scratch_load"), "scratch_store", g("scratch_store"), "global", g("global_"), "ds", g("ds_"), "s_load", g("s_load"))
print("  v_readlane", c["v_readlane_b32"], "v_writelane", c["v_writelane_b32"], "v_pk", g("v_pk_"), "v_cndmask", c["v_cndmask_b32_e32"] + c["v_cndmask_b32_e64"], "v_mov", c["v_mov_b32_e32"], "s_cbranch", g("s_cbranch"))
if len(sys.argv) > 3: print("  top:", c.most_common(int(sys.argv[3])))
