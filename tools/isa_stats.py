"""dev helper: per-kernel ISA statistics of the gfx950 code object (run after `make` in csrc)"""
import collections
import re
import subprocess
import sys

SRC = sys.argv[1] if len(sys.argv) > 1 else "watermarking-gpu_amd/csrc/wm_kernels.hip"
FILT = sys.argv[2:] or ["k_gramIfE", "k_me_statsIfE", "k_embedIffLi1ELi0", "k_detectIfLi0", "k_embedIhhLi1ELi0", "k_detectIhLi0"]
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize", "--offload-arch=gfx950",
                       "--cuda-device-only", "-S", SRC, "-o", "/tmp/wm_kernels.s"])
s = open("/tmp/wm_kernels.s").read()
meta = {}
for m in re.finditer(r"\.group_segment_fixed_size:\s+(\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+)", s, re.S):
    meta[m.group(2)] = dict(lds=int(m.group(1)), scratch=int(m.group(3)), sgpr=int(m.group(4)), vgpr=int(m.group(5)))
for name, md in meta.items():
    if not any(f in name for f in FILT):
        continue
    i = s.index("\n" + name + ":")
    k = s.index(".Lfunc_end", i)
    ins = [l.strip() for l in s[i:k].split("\n") if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    c = collections.Counter()
    for t in ins:
        op = t.split()[0]
        if op.startswith("global_load") or op.startswith("buffer_load"): c["gload"] += 1
        elif op.startswith("global_store"): c["gstore"] += 1
        elif op.startswith("ds_"): c["lds"] += 1
        elif op.startswith("s_waitcnt"):
            c["wait"] += 1
            if "vmcnt(0)" in t: c["vmcnt0"] += 1
        elif op.startswith("v_mov") or op.startswith("v_accvgpr"): c["vmov"] += 1
        elif op.startswith("v_"): c["valu"] += 1
        elif op.startswith("s_cbranch") or op.startswith("s_branch"): c["branch"] += 1
        elif op.startswith("s_and_saveexec") or op.startswith("s_or_saveexec"): c["saveexec"] += 1
        elif op.startswith("scratch_"): c["scratch"] += 1
        else: c["salu"] += 1
    print(f"{name[:70]:70s} vgpr={md['vgpr']:3d} sgpr={md['sgpr']:3d} lds={md['lds']:5d} scr={md['scratch']} n={len(ins):5d} {dict(c)}")
