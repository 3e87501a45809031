"""dev helper: for one kernel, print the biggest loop blocks with their instruction mix and waitcnts"""
import collections, re, subprocess, sys
src, sym = sys.argv[1], sys.argv[2]
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize", "--offload-arch=gfx950"] + sys.argv[4:] + [
                       "--cuda-device-only", "-S", src, "-o", "/tmp/k.s"], stderr=subprocess.DEVNULL)
s = open("/tmp/k.s").read()
names = [m.group(1) for m in re.finditer(r"^(_ZN3wmk\S+):", s, re.M) if sym in m.group(1)]
for nm in names[:1]:
    i = s.index("\n" + nm + ":"); k = s.index(".Lfunc_end", i)
    blocks = []; cur = ["entry", "", []]; blocks.append(cur)
    for l in s[i:k].split("\n"):
        m = re.match(r"^(\.LBB\d+_\d+):(.*)", l)
        if m:
            cur = [m.group(1), m.group(2), []]; blocks.append(cur)
        elif l.startswith("\t") and not l.strip().startswith((".", ";")):
            cur[2].append(l.strip())
    md = re.search(r"\.name:\s+" + re.escape(nm) + r".*?\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+)", s, re.S)
    print(nm[:70], "sgpr", md.group(1), "vgpr", md.group(2), "blocks", len(blocks))
    big = sorted(blocks, key=lambda b: -len(b[2]))[:int(sys.argv[3]) if len(sys.argv) > 3 else 3]
    for lab, note, ins in big:
        c = collections.Counter(t.split()[0] for t in ins)
        valu = sum(v for k2, v in c.items() if k2.startswith("v_"))
        print(f"  {lab} {'LOOP' if 'Loop' in note else ''} n={len(ins)} valu={valu} salu={sum(v for k2,v in c.items() if k2.startswith('s_') and not k2.startswith('s_waitcnt'))} "
              f"loads={sum(v for k2,v in c.items() if 'load' in k2)} stores={sum(v for k2,v in c.items() if 'store' in k2)} lds={sum(v for k2,v in c.items() if k2.startswith('ds_'))} vmov={sum(v for k2,v in c.items() if k2.startswith('v_mov'))}")
        print("    waits:", [t.replace("s_waitcnt ", "") for t in ins if t.startswith("s_waitcnt")])
