"""dev measurement: PCIe-inclusive frame rate of the video path (u8 Y planes in pinned host memory, WM_MEM_HOST
planes, several slots in flight).  Reported in DESIGN.md; never the bench.py `value`."""
import ctypes as C
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
wm = importlib.import_module("watermarking-gpu_amd")
synth = importlib.import_module("watermarking-gpu_amd.synth")


def main(R=2160, Cc=3840, slots=4, frames=240, detect=True, F=1):
    L = wm.lib()
    W = synth.synth_watermark(R, Cc)
    eng = wm.Watermark(R, Cc, W, 3, 40.0, nslots=slots, max_frames=F)
    n = R * Cc
    src = synth.synth_frame(R, Cc, frame=0, dtype=np.uint8)
    ins, outs, planes_in, planes_out = [], [], [], []
    for s in range(slots):
        pi, po = L.wm_host_alloc(n * F), L.wm_host_alloc(n * F)
        hi = np.ctypeslib.as_array(C.cast(pi, C.POINTER(C.c_uint8)), shape=(F, R, Cc))
        hi[:] = src
        ins.append(pi); outs.append(po)
        planes_in.append(wm.wm_plane(pi, R, Cc, 1, wm.WM_U8, wm.WM_MEM_HOST, F, Cc, 0, n))
        planes_out.append(wm.wm_plane(po, R, Cc, 1, wm.WM_U8, wm.WM_MEM_HOST, F, Cc, 0, n))
    a = [(C.c_float * F)() for _ in range(slots)]
    corr = [(C.c_float * F)() for _ in range(slots)]

    def run(nf):
        for f in range(nf):
            s = f % slots
            if f >= slots:
                L.wm_sync(eng._ctx, s)
            L.wm_embed(eng._ctx, 0, C.byref(planes_in[s]), C.byref(planes_in[s]), C.byref(planes_out[s]), a[s], None, s)
            if detect:
                L.wm_detect(eng._ctx, 0, C.byref(planes_out[s]), corr[s], None, s)
        for s in range(slots):
            L.wm_sync(eng._ctx, s)
    run(2 * slots)
    t0 = time.perf_counter()
    run(frames // F)
    dt = time.perf_counter() - t0
    frames = (frames // F) * F
    mb = (2 + (1 if detect else 0)) * n / 1e6
    print(f"{Cc}x{R} u8 host-staged, {F} frames per call, slots={slots}, {'embed+detect' if detect else 'embed only'}: {frames / dt:8.1f} frames/s, "
          f"{frames * mb / dt / 1e3:6.2f} GB/s over PCIe (H2D+D2H), a={a[0][0]:.4f} corr={corr[0][0]:.5f}", flush=True)
    eng.close()


if __name__ == "__main__":
    for F, s in ((1, 2), (1, 8), (4, 2), (4, 3), (8, 2), (8, 3), (16, 3)):
        main(slots=s, detect=False, F=F, frames=480)
    for F, s in ((1, 4), (8, 3)):
        main(slots=s, detect=True, F=F, frames=480)
