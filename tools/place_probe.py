"""dev helper (GPU box): does k_embed's time depend on WHERE the output plane lies relative to the input plane?
(one allocation, y carved out at x_end + delta for several deltas; per-kernel events of 8 serial steps each)"""
import ctypes as C
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
wm = importlib.import_module("watermarking-gpu_amd")
from quick_bench import fake_frames  # noqa: E402

R, Cc, F = 2160, 3840, 16
N = R * Cc
W = torch.randn((R, Cc), generator=torch.Generator().manual_seed(2)).numpy()
eng = wm.Watermark(R, Cc, W, 3, 40.0, nslots=1, max_frames=F)
src = fake_frames(R, Cc, F)
big = torch.empty(2 * F * N + (64 << 20), dtype=torch.float32, device="cuda")
print("base address mod 2 MiB:", big.data_ptr() % (2 << 20), flush=True)
x = big[:F * N].view(F, R, Cc)
x.copy_(src)
a, corr = (C.c_float * F)(), (C.c_float * F)()
for delta in (0, 64, 1024, 4096, 16384, 65536, 262144, 1 << 20, (1 << 20) + 4096, 3 << 20, 5 << 20, (8 << 20) + 1024, 12 << 20):   # elements
    y = big[F * N + delta: F * N + delta + F * N].view(F, R, Cc)
    px, py = wm.plane_of(x), wm.plane_of(y)
    for _ in range(2):
        eng.embed_async(px, px, py, 0, 0, a_out=a); eng.detect_async(py, 0, 0, corr_out=corr); eng.sync(0)
    eng.prof_enable(True); eng.prof_reset()
    for _ in range(8):
        eng.embed_async(px, px, py, 0, 0, a_out=a); eng.detect_async(py, 0, 0, corr_out=corr)
    eng.sync(0)
    rep = eng.prof_report(); eng.prof_enable(False)
    print(f"delta {4 * delta:>10d} B  (y - x) mod 1 MiB = {(4 * (F * N + delta)) % (1 << 20):>8d}: " +
          "  ".join(f"{k}:{1e3 * ms / n:.1f}" for k, (n, ms) in rep.items()), flush=True)
eng.close()
