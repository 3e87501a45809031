"""ad-hoc timing helper for development runs on the GPU box (not part of the test suite)"""
import ctypes as C
import importlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
wm = importlib.import_module("watermarking-gpu_amd")
if os.environ.get("WM_AB_LIB"):  # development A/B runs: another build of the library (tools/ab.py)
    wm.LIB_PATH = os.environ["WM_AB_LIB"]
    _L = C.CDLL(wm.LIB_PATH)   # an older build may lack entries added since: bind what it has
    wm.ABI = [e for e in wm.ABI if hasattr(_L, e[0])]
synth = importlib.import_module("watermarking-gpu_amd.synth")


def fake_frames(rows, cols, F, dtype=torch.float32):
    g = torch.Generator(device="cuda").manual_seed(1)
    r = torch.arange(rows, device="cuda", dtype=torch.float32)[:, None]
    c = torch.arange(cols, device="cuda", dtype=torch.float32)[None, :]
    base = 128 + 56 * torch.sin(2 * np.pi * r / 97) * torch.cos(2 * np.pi * c / 61) + 36 * torch.sin(2 * np.pi * (r + 2 * c) / 389)
    out = []
    for f in range(F):
        n = torch.randn((rows + 3, cols + 3), device="cuda", generator=g)
        n1 = torch.nn.functional.avg_pool2d(n[None, None], 4, stride=1)[0, 0] * 4
        x = (base + 24 * n1 + 6 * torch.randn((rows, cols), device="cuda", generator=g)).clamp(0, 255)
        out.append(x.to(dtype) if dtype == torch.float32 else x.round().to(torch.uint8))
    return torch.stack(out)


def run(rows, cols, F, nslots, iters, dtype=torch.float32, rps=0, mask=0, p=3):
    W = torch.randn((rows, cols), generator=torch.Generator().manual_seed(2)).numpy()
    eng = wm.Watermark(rows, cols, W, p, 40.0, nslots=nslots, max_frames=F)
    if rps:
        eng.set_rows_per_segment(rps)
    xs = [fake_frames(rows, cols, F, dtype) for _ in range(nslots)]
    ys = [torch.empty_like(x) for x in xs]
    torch.cuda.synchronize()
    a = [(C.c_float * F)() for _ in range(nslots)]
    corr = [(C.c_float * F)() for _ in range(nslots)]

    px = [wm.plane_of(x) for x in xs]
    py = [wm.plane_of(y) for y in ys]
    if os.environ.get("WM_QB_HANDOVER"):  # Gram hand-over: the detector reads the slot's last embed output
        eng.set_handover(True)
        pd = wm.wm_plane(None, rows, cols, 1, py[0].dtype, wm.WM_MEM_SLOT_OUT, F, cols, 0, rows * cols)
        py_det = [pd for _ in range(nslots)]
    else:
        py_det = py

    def step():
        for s in range(nslots):
            eng.embed_async(px[s], px[s], py[s], mask, s, a_out=a[s])
            eng.detect_async(py_det[s], mask, s, corr_out=corr[s])
        for s in range(nslots):
            eng.sync(s)
    for _ in range(3):
        step()
    t0 = time.perf_counter()
    for _ in range(iters):
        step()
    dt = time.perf_counter() - t0
    fps = iters * nslots * F / dt
    N = rows * cols
    es = 4 if dtype == torch.float32 else 1
    alg = (4 + 4 + 4) * N + 2 * es * N * 3 if False else None
    bytes_frame = (36 if dtype == torch.float32 else 18) * N
    print(f"{rows}x{cols} {str(dtype)[6:]} F={F} slots={nslots} rps={rps}{'' if p == 3 else ' p=%d' % p}: {fps:9.1f} frames/s  {1e6 * dt / (iters * nslots * F):8.1f} us/frame  "
          f"{fps * bytes_frame / 1e12:6.3f} TB/s algorithmic  a={a[0][0]:.4f} corr={corr[0][0]:.5f}", flush=True)
    eng.prof_enable(True)
    eng.prof_reset()
    for _ in range(5):
        step()
    rep = eng.prof_report()
    eng.prof_enable(False)
    print("   " + "  ".join(f"{k}:{1e3 * ms / n / F:.1f}us" for k, (n, ms) in rep.items()), flush=True)
    eng.close()
    return fps


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "4k"
    if which == "4k":
        for F, ns in ((1, 1), (1, 2), (1, 4), (4, 2), (8, 2), (16, 1)):
            run(2160, 3840, F, ns, 50)
        for rps in (8, 16, 24, 32, 48):
            run(2160, 3840, 4, 2, 30, rps=rps)
        run(2160, 3840, 8, 2, 30, dtype=torch.uint8)
        run(1080, 1920, 8, 2, 50)
        run(4320, 7680, 2, 2, 10)
