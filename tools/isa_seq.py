"""dev helper: print load / wait / branch positions of the loop blocks of a kernel that contain `marker`"""
import re, subprocess, sys
src, sym, marker = sys.argv[1], sys.argv[2], sys.argv[3]
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950",
                       "--cuda-device-only", "-S", src, "-o", "/tmp/k.s"], stderr=subprocess.DEVNULL)
s = open("/tmp/k.s").read()
nm = [m.group(1) for m in re.finditer(r"^(_ZN3wmk\S+):", s, re.M) if sym in m.group(1)][0]
i0 = s.index("\n" + nm + ":"); k0 = s.index(".Lfunc_end", i0)
for m in re.finditer(r"^(\.LBB\d+_\d+):.*Loop", s[i0:k0], re.M):
    blk = s[i0 + m.start():k0].split("\n"); n = 0; out = []; has = False
    for l in blk[1:]:
        t = l.strip()
        if re.match(r"^\.LBB", l): break
        if not l.startswith("\t") or t.startswith((".", ";")): continue
        n += 1
        if marker in t: has = True
        if t.startswith(("global_load", "global_store", "s_waitcnt", "s_cbranch")): out.append(f"{n}:{t.split()[0].replace('global_','g_')}{' '+t.split()[1] if 'waitcnt' in t else ''}")
    if has and n > 200:
        print(m.group(1), "n =", n); print("  " + "  ".join(out))
