"""dev helper (GPU box): one 4K frame per call, the reference's call pattern (synchronous makeWatermark, then
synchronous detectWatermark), timed on the host; run under rocprofv3 --kernel-trace for kernel durations and gaps.
usage: python tools/f1_trace.py [iters] [dtype] [rows cols]"""
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

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dtype = torch.uint8 if len(sys.argv) > 2 and sys.argv[2] == "u8" else torch.float32
R, Cc = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (2160, 3840)
W = torch.randn((R, Cc), generator=torch.Generator().manual_seed(2)).numpy()
eng = wm.Watermark(R, Cc, W, 3, 40.0)
x = fake_frames(R, Cc, 1, dtype)[0].contiguous()
y = torch.empty_like(x)
px, py = wm.plane_of(x), wm.plane_of(y)
a, corr = (C.c_float * 1)(), (C.c_float * 1)()
L = wm.lib()
for mask in (0, 1):
    def embed():
        return L.wm_embed(eng._ctx, mask, C.byref(px), C.byref(px), C.byref(py), a, None, wm.WM_SLOT_SYNC)

    def detect():
        return L.wm_detect(eng._ctx, mask, C.byref(py), corr, None, wm.WM_SLOT_SYNC)
    for _ in range(5):
        embed(); detect()
    t0 = time.perf_counter()
    for _ in range(iters):
        embed()
    t1 = time.perf_counter()
    for _ in range(iters):
        detect()
    t2 = time.perf_counter()
    for _ in range(iters):
        embed(); detect()
    t3 = time.perf_counter()
    print(f"{R}x{Cc} {str(dtype)[6:]} mask={mask} sync calls: embed {1e6*(t1-t0)/iters:.1f} us  detect {1e6*(t2-t1)/iters:.1f} us  "
          f"pair {1e6*(t3-t2)/iters:.1f} us  a={a[0]:.4f} corr={corr[0]:.5f}", flush=True)
eng.close()
