"""dev helper (GPU box): A/B timing of two builds of libwm_hip.so in alternating child processes
usage: python tools/ab.py libA.so libB.so [libC.so ...] [dtype]   (paths relative to the repo root)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = [os.path.join(ROOT, p) for p in sys.argv[1:] if p.endswith(".so")]
dt = sys.argv[-1] if sys.argv[-1] in ("f32", "u8") else "f32"
code = ("import sys, torch; sys.path.insert(0, 'tools'); from quick_bench import run; "
        "dt = torch.uint8 if '%s' == 'u8' else torch.float32; " % dt)
shapes = os.environ.get("WM_AB_SHAPES", "16x1,16x3")   # FxS list
mask = int(os.environ.get("WM_AB_MASK", "0"))          # 0 ME, 1 NVF
rows, cols = (int(v) for v in os.environ.get("WM_AB_SIZE", "2160x3840").split("x"))
code += "".join("run(%d, %d, %s, %s, %d, dtype=dt, mask=%d); " % (rows, cols, fs.split("x")[0], fs.split("x")[1], max(20, 100 // int(fs.split("x")[0])), mask) for fs in shapes.split(","))
for rep in range(int(os.environ.get('WM_AB_REPS', '2'))):
    for lib in libs:
        env = dict(os.environ, WM_AB_LIB=lib)
        out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True).stdout
        print("==", os.path.basename(lib), "rep", rep)
        print("\n".join(l for l in out.splitlines() if "frames/s" in l or "k_gram" in l), flush=True)
