"""dev helper (GPU box): soak of the Gram hand-over leg -- the bench loop (3 slots x 16 4K frames, detector on WM_MEM_SLOT_OUT)
for many steps; every step's scores and strengths must equal the first step's bit for bit (fold tails, tickets, record arrays)"""
import ctypes as C
import importlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
wm = importlib.import_module("watermarking-gpu_amd")
from quick_bench import fake_frames  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
R, Cc, F, S = 2160, 3840, 16, 3
W = torch.randn((R, Cc), generator=torch.Generator().manual_seed(2)).numpy()
eng = wm.Watermark(R, Cc, W, 3, 40.0, nslots=S, max_frames=F)
eng.set_handover(True)
xs = [fake_frames(R, Cc, F) for _ in range(S)]
ys = [torch.empty_like(x) for x in xs]
px, py = [wm.plane_of(x) for x in xs], [wm.plane_of(y) for y in ys]
sp = wm.wm_plane(None, R, Cc, 1, wm.WM_F32, wm.WM_MEM_SLOT_OUT, F, Cc, 0, R * Cc)
a = [(C.c_float * F)() for _ in range(S)]
c = [(C.c_float * F)() for _ in range(S)]
first, bad = None, 0
t0 = time.perf_counter()
for it in range(steps):
    mask = it % 2 if it % 7 == 3 else 0     # now and then an NVF step in between (other kernels through the same arrays)
    for s in range(S):
        eng.embed_async(px[s], px[s], py[s], mask, s, a_out=a[s])
        eng.detect_async(sp, mask, s, corr_out=c[s])
    for s in range(S):
        eng.sync(s)
    if mask == 0:
        got = [list(v) for v in a] + [list(v) for v in c]
        if first is None:
            first = got
        elif got != first:
            bad += 1
            print("step", it, "differs", flush=True)
print(f"{steps} steps in {time.perf_counter() - t0:.1f} s, {bad} differing steps", flush=True)
eng.close()
sys.exit(1 if bad else 0)
