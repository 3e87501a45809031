"""dev helper: instruction mix of a kernel's loops (whole back-edge ranges, conditional blocks inside included)
usage: python tools/isa_loopmix.py file.s <mangled-name-substring> [n loops]"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
sym = sys.argv[2]
nm = [m.group(1) for m in re.finditer(r"^(_Z\S+):", s, re.M) if sym in m.group(1)][0]
i = s.index("\n" + nm + ":"); k = s.index(".Lfunc_end", i)
lines = s[i:k].split("\n")
lab = {}
for n, l in enumerate(lines):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        lab[m.group(1)] = n
loops = []
for n, l in enumerate(lines):
    m = re.search(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in lab and lab[m.group(1)] < n:
        loops.append((lab[m.group(1)], n))
md = re.search(r"\.name:\s+" + re.escape(nm) + r".*?\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+)", s, re.S)
print(nm[:60], "sgpr", md.group(1), "vgpr", md.group(2))
for a, b in sorted(loops, key=lambda t: -(t[1] - t[0]))[:int(sys.argv[3]) if len(sys.argv) > 3 else 3]:
    ins = [l.strip() for l in lines[a:b + 1] if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    c = collections.Counter(t.split()[0] for t in ins)
    print(f"lines {a}-{b}: n={len(ins)} valu={sum(v for k2, v in c.items() if k2.startswith('v_'))} salu={sum(v for k2, v in c.items() if k2.startswith('s_'))} "
          f"loads={sum(v for k2, v in c.items() if 'load' in k2)}")
    print("   ", sorted(((v, k2) for k2, v in c.items() if k2.startswith("v_")), reverse=True)[:16])
    print("    waits", [t.replace("s_waitcnt ", "") for t in ins if t.startswith("s_waitcnt")])
