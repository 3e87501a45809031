"""dev helper: VGPRs live across the back-edge of a kernel's biggest loop (read before written in the body), and the
registers each global/buffer load of the body writes.  usage: python tools/isa_livein.py file.s <mangled-substring> [loop rank]"""
import re
import sys

s = open(sys.argv[1]).read()
nm = [m.group(1) for m in re.finditer(r"^(_Z\S+):", s, re.M) if sys.argv[2] in m.group(1)][0]
i = s.index("\n" + nm + ":"); k = s.index(".Lfunc_end", i)
lines = s[i:k].split("\n")
lab = {m.group(1): n for n, l in enumerate(lines) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
loops = []
for n, l in enumerate(lines):
    m = re.search(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in lab and lab[m.group(1)] < n:
        loops.append((lab[m.group(1)], n))
loops.sort(key=lambda t: -(t[1] - t[0]))
a, b = loops[int(sys.argv[3]) if len(sys.argv) > 3 else 0]


def regs(tok):
    out = []
    for m in re.finditer(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]", tok):
        if m.group(1):
            out.append(int(m.group(1)))
        else:
            out += list(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


written, livein = set(), set()
for l in lines[a:b + 1]:
    t = l.strip()
    if not l.startswith("\t") or t.startswith((".", ";", "s_")):
        continue
    t = t.split(";")[0]
    parts = t.split(None, 1)
    if len(parts) < 2:
        continue
    ops = [o.strip() for o in parts[1].split(",")]
    op = parts[0]
    nd = 0 if "store" in op else 1
    for o in ops[nd:]:
        for r in regs(o):
            if r not in written:
                livein.add(r)
    if op.startswith("v_fmac") or op.startswith("v_mov_b32_dpp") or "dpp" in t:
        for r in regs(ops[0]):
            if r not in written:
                livein.add(r)
    for r in regs(ops[0]) if nd else []:
        written.add(r)
print(f"loop lines {a}-{b}: {len(livein)} VGPRs live across the back-edge, {len(written)} written in the body")
