"""dev helper (GPU box): k_detect's execution window by in-kernel stamps against the duration its launch events report
(build: tools/build_variant.sh kstamp "-DWM_KSTAMP"; run: WM_AB_LIB=watermarking-gpu_amd/libwm_ab_kstamp.so python tools/kstamp.py)"""
import ctypes as C
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
wm = importlib.import_module("watermarking-gpu_amd")
wm.LIB_PATH = os.path.join(ROOT, os.environ["WM_AB_LIB"])
from quick_bench import fake_frames  # noqa: E402

L = wm.lib()
L.wm_dbg_kstamp.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
R, Cc, F = 2160, 3840, 16
W = torch.randn((R, Cc), generator=torch.Generator().manual_seed(2)).numpy()
eng = wm.Watermark(R, Cc, W, 3, 40.0, nslots=1, max_frames=F)
x = fake_frames(R, Cc, F)
y = torch.empty_like(x)
px, py = wm.plane_of(x), wm.plane_of(y)
a, corr = (C.c_float * F)(), (C.c_float * F)()
eng.embed_async(px, px, py, 0, 0, a_out=a); eng.detect_async(py, 0, 0, corr_out=corr); eng.sync(0)
eng.prof_enable(True)
st = (C.c_ulonglong * 2)()
for mode in ("detect alone, sync per call", "embed + detect back to back"):
    wins, evs = [], []
    for it in range(12):
        L.wm_dbg_kstamp(None, 1)
        eng.prof_reset()
        if mode.startswith("embed"):
            eng.embed_async(px, px, py, 0, 0, a_out=a)
        eng.detect_async(py, 0, 0, corr_out=corr)
        eng.sync(0)
        torch.cuda.synchronize()
        L.wm_dbg_kstamp(st, 0)
        rep = eng.prof_report()
        wins.append((st[1] - st[0]) / 100.0)
        evs.append(1e3 * rep["k_detect"][1] / rep["k_detect"][0])
    wins, evs = sorted(wins[2:]), sorted(evs[2:])
    print(f"{mode}: waves execute over {wins[len(wins)//2]:.1f} us (median; min {wins[0]:.1f}), the launch's events say {evs[len(evs)//2]:.1f} us (min {evs[0]:.1f})")
eng.close()
