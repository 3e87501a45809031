"""dev helper (GPU box): do the compute-dense sweeps slow down as the board reaches its power limit?  The serialised 16-frame 4K
pipeline for ~4 s; every 0.4 s: per-kernel event times of the last steps, engine clock and power (amdgpu sysfs)"""
import ctypes as C
import glob
import importlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
wm = importlib.import_module("watermarking-gpu_amd")
from quick_bench import fake_frames  # noqa: E402


def rd(pat):
    vals = []
    for p in glob.glob(pat):
        try:
            vals.append(float(open(p).read()))
        except Exception:
            pass
    return max(vals) if vals else float("nan")


R, Cc, F, S = 2160, 3840, 16, int(sys.argv[1]) if len(sys.argv) > 1 else 1
W = torch.randn((R, Cc), generator=torch.Generator().manual_seed(2)).numpy()
eng = wm.Watermark(R, Cc, W, 3, 40.0, nslots=S, max_frames=F)
xs = [fake_frames(R, Cc, F) for _ in range(S)]
ys = [torch.empty_like(x) for x in xs]
px, py = [wm.plane_of(x) for x in xs], [wm.plane_of(y) for y in ys]
a, corr = (C.c_float * F)(), (C.c_float * F)()
time.sleep(1.0)  # start from an idle board
t_end = time.perf_counter() + 4.0
k = 0
while time.perf_counter() < t_end:
    # load phase: S slots in flight, no events
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < 0.4:
        for s in range(S):
            eng.sync(s)
            eng.embed_async(px[s], px[s], py[s], 0, s, a_out=a); eng.detect_async(py[s], 0, s, corr_out=corr)
        n += S
    for s in range(S):
        eng.sync(s)
    fps = n * F / (time.perf_counter() - t0)
    clk, pw = rd("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input") / 1e6, rd("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average") / 1e6
    # probe: 3 serial steps on slot 0 with events attached
    eng.prof_enable(True); eng.prof_reset()
    for _ in range(3):
        eng.embed_async(px[0], px[0], py[0], 0, 0, a_out=a); eng.detect_async(py[0], 0, 0, corr_out=corr)
    eng.sync(0)
    rep = eng.prof_report(); eng.prof_enable(False)
    print(f"t={0.4 * (k + 1):.1f}s  {fps:8.0f} frames/s  sclk {clk:5.0f} MHz  {pw:6.0f} W | " + "  ".join(f"{kn}:{1e3 * ms / cnt:.1f}us" for kn, (cnt, ms) in rep.items()), flush=True)
    k += 1
eng.close()
