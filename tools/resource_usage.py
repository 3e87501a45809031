"""dev helper: registers, spills, scratch and occupancy of every kernel instance (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/resource_usage.py [file.hip ...] [--spills-only] [extra -D flags]
Exit code 1 when any instance spills VGPRs or uses scratch (the check `make resource-usage` is for)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "watermarking-gpu_amd", "csrc")
files = [a for a in sys.argv[1:] if a.endswith(".hip")] or ["wm_k_gram.hip", "wm_k_embed.hip", "wm_k_detect.hip", "wm_k_fused.hip"]
flags = [a for a in sys.argv[1:] if a.startswith("-D")]
only = "--spills-only" in sys.argv
bad = 0
for f in files:
    p = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize", "--offload-arch=gfx950",
                        "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(CSRC, f), "-o", "/dev/null"] + flags, capture_output=True, text=True)
    cur = None
    recs = []
    for ln in p.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", ln)
        if m:
            cur = {"name": m.group(1)}
            recs.append(cur)
            continue
        for key, pat in (("vgpr", r"\bVGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("sgpr", r"\bSGPRs: (\d+)"), ("vspill", r"VGPRs Spill: (\d+)"),
                         ("sspill", r"SGPRs Spill: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"),
                         ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, ln)
            if m and cur is not None:
                cur[key] = int(m.group(1))
    for r in recs:
        dem = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
        short = dem.split("(")[0].replace("void wmk::", "")
        spills = r.get("vspill", 0) > 0 or r.get("scratch", 0) > 0
        # (k_fused_pair's 128-row instances: 4-8 VGPRs stored and reloaded ONCE per wave around the statistics hand-off -- the
        # opt-in one-launch pair, WM_FUSED_PAIR=1; reported, not counted)
        bad += spills and "k_fused_pair" not in short
        if only and not spills:
            continue
        print(f"{short[:84]:84s} vgpr {r.get('vgpr', 0):3d} sgpr {r.get('sgpr', 0):3d} vspill {r.get('vspill', 0):3d} sspill {r.get('sspill', 0):3d} "
              f"scratch {r.get('scratch', 0):4d} lds {r.get('lds', 0):6d} waves/SIMD {r.get('occ', 0)}")
sys.exit(1 if bad else 0)
