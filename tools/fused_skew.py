"""dev helper (GPU box): which workgroups of a fused launch arrive last at the first hand-off?  Per-workgroup start and
record-ready stamps (WM_FUSED_STAMPS) of a 4K f32 ME embed, averaged over calls, by workgroup index / XCD / row band."""
import ctypes as C
import importlib
import os
import sys

import numpy as np
import torch

os.environ["WM_FUSED_STAMPS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
wm = importlib.import_module("watermarking-gpu_amd")
from quick_bench import fake_frames  # noqa: E402

R, Cc = 2160, 3840
W = torch.randn((R, Cc), generator=torch.Generator().manual_seed(2)).numpy()
eng = wm.Watermark(R, Cc, W, 3, 40.0)
act, G, th, fb = eng.fused_info()
x = fake_frames(R, Cc, 1)[0].contiguous()
acc = []
for it in range(40):
    y, a = eng.makeWatermark(x, x, 0)
    torch.cuda.synchronize()
    if it >= 5:
        buf = (C.c_ulonglong * (G * 16 + 16))()
        n = wm.lib().wm_fused_stamps(eng._ctx, buf, G * 16 + 16)
        st = np.array(buf[:n], dtype=np.float64).reshape(-1, 16)[:-1]
        acc.append((st - st[:, 0].min()) / 100.0)
m = np.mean(acc, axis=0)
start, first8, ready, ticket = m[:, 0], m[:, 12], m[:, 1], m[:, 2]
ids = np.arange(G)
ns = 15
print("workgroups", G, " start: median %.2f max %.2f   record ready: median %.2f max %.2f" % (np.median(start), start.max(), np.median(ready), ready.max()))
print("corr(start, ready) = %.2f   mean(ready - start) = %.2f +- %.2f" % (np.corrcoef(start, ready)[0, 1], (ready - start).mean(), (ready - start).std()))
print("by XCD (id % 8):   start / ready / ready-start")
for xcd in range(8):
    s = ids % 8 == xcd
    print("   xcd %d: %.2f  %.2f  %.2f" % (xcd, start[s].mean(), ready[s].mean(), (ready - start)[s].mean()))
print("by row band (id // 15):")
for b in range((G + ns - 1) // ns):
    s = ids // ns == b
    print("   band %2d: start %.2f  ready %.2f  dur %.2f" % (b, start[s].mean(), ready[s].mean(), (ready - start)[s].mean()))
print("by strip (id % 15):")
for b in range(ns):
    s = ids % ns == b
    print("   strip %2d: start %.2f  ready %.2f  dur %.2f" % (b, start[s].mean(), ready[s].mean(), (ready - start)[s].mean()))
order = np.argsort(-ready)[:12]
print("last 12 workgroups:", [(int(i), round(float(start[i]), 2), round(float(ready[i]), 2)) for i in order])
