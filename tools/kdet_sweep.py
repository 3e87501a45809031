"""dev helper (GPU box): per-kernel times of the serialised 16-frame 4K pipeline for several builds and segment lengths
usage: python tools/kdet_sweep.py lib1.so [lib2.so ...] [f32|u8] -- rps list from WM_SWEEP_RPS (default "0")"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = [p for p in sys.argv[1:] if p.endswith(".so")]
dt = sys.argv[-1] if sys.argv[-1] in ("f32", "u8") else "f32"
rps = os.environ.get("WM_SWEEP_RPS", "0").split(",")
code = ("import sys, torch; sys.path.insert(0, 'tools'); from quick_bench import run; dt = torch.uint8 if '%s' == 'u8' else torch.float32; " % dt
        + "".join("run(2160, 3840, 16, 1, 12, dtype=dt, rps=%s); " % r for r in rps))
for lib in libs:
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=dict(os.environ, WM_AB_LIB=os.path.join(ROOT, lib)), capture_output=True, text=True)
    print("==", os.path.basename(lib), flush=True)
    lines = out.stdout.splitlines()
    for a, b in zip(lines[0::2], lines[1::2]):
        print("  ", a.split(":")[0].split("rps=")[1], a.split(":")[1].split("frames/s")[0].strip(), "f/s |", b.strip(), flush=True)
    if out.returncode:
        print(out.stderr[-800:])
