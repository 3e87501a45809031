"""dev helper (GPU box): phase time line of the fused single-frame kernels from their in-kernel stamps
usage: WM_FUSED_STAMPS=1 python tools/fused_stamps.py [rows cols] [dtype]"""
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
from quick_bench import fake_frames  # noqa: E402  (quick_bench honours WM_AB_LIB: another build of the library)

R, Cc = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2160, 3840)
dtype = torch.uint8 if len(sys.argv) > 3 and sys.argv[3] == "u8" else torch.float32
W = torch.randn((R, Cc), generator=torch.Generator().manual_seed(2)).numpy()
eng = wm.Watermark(R, Cc, W, 3, 40.0)
act, G, th, fb = eng.fused_info()
print(f"{R}x{Cc}: fused={act} workgroups={G} tile_rows={th}")
x = fake_frames(R, Cc, 1, dtype)[0].contiguous()
NAMES = ["start", "record ready", "ticket A", "coef known", "phase B done", "ticket B", "scalars known", "end",
         "border parked (w0)", "first row in (w0)", "march done (w0)", "rows requested (w8)", "first row in (w8)", "march done (w8)",
         "at barrier (w8)", "at barrier (w0)"]
ORDER = [0, 11, 8, 9, 12, 10, 13, 14, 15, 1, 2, 3, 4, 5, 6, 7]


def stamps():
    buf = (C.c_ulonglong * (G * 16 + 16))()
    n = wm.lib().wm_fused_stamps(eng._ctx, buf, G * 16 + 16)
    return np.array(buf[:n], dtype=np.float64).reshape(-1, 16)


for mask in (0, 1):
    for op in ("embed", "detect"):
        acc = []
        for it in range(12):
            if op == "embed":
                y, a = eng.makeWatermark(x, x, mask)
            else:
                eng.detectWatermark(y, mask)
            torch.cuda.synchronize()
            if it >= 2:
                st = stamps()
                t0 = st[:-1, 0].min()
                acc.append((st - t0) / 100.0)  # us; last row: the folding workgroup's extra stamps
        m = np.mean(acc, axis=0)  # [G + 1][8]
        extra, m = m[-1], m[:-1]
        if op == "detect" or mask == 0:
            print(f"   (folding workgroup: records folded {extra[0]:.2f}, totals ready {extra[1]:.2f}, solved + published {extra[2]:.2f})")
        print(f"mask={mask} {op}: per phase boundary, us after the first workgroup's start: min / median / max over workgroups")
        for k in ORDER:
            col = m[:, k]
            if (col <= 0).all() and k:
                continue
            nz = col[col > 0] if k else col
            if nz.size:
                print(f"   {NAMES[k]:20s} {nz.min():7.2f} {np.median(nz):7.2f} {nz.max():7.2f}")
eng.close()
