"""Stress tests of the in-kernel hand-offs (the fold tails of the sweeps, the tickets / granules of the fused
kernels): what tools/soak.py and tools/fuzz.py run at length, bounded so that the whole file takes well under a minute.
Every result must be BITWISE equal to the first result of the same input: a fold that reads a stale partial record, a
ticket or counter left non-zero, a granule of an earlier call taken for this call's, or a race between slots shows up
as a changed strength / correlation / output checksum."""
import ctypes as C
import importlib
import os
import sys
import threading

import numpy as np
import pytest

from synth import synth_frame, synth_watermark

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def tc():
    import torch
    assert torch.cuda.is_available()
    return torch


def test_soak_batched_pipeline(wm, tc):
    """alternating input batches on 3-4 slots, frame-quad launches, 200+ steps per configuration"""
    soak = importlib.import_module("soak")
    torch = tc
    assert soak.soak(2160, 3840, 16, 3, 200, torch.float32, 0) == 0
    assert soak.soak(1080, 1920, 8, 4, 200, torch.uint8, 1) == 0
    assert soak.soak(300, 700, 5, 4, 400, torch.float32, 0) == 0


def test_fuzz_slice(wm, tc):
    """60 seeded random shapes / element types / masks / window sizes / segment lengths / batch sizes against the oracle"""
    fuzz = importlib.import_module("fuzz")
    assert fuzz.run(60, seed=11) == 0


@pytest.mark.parametrize("shape", [(2160, 3840), (1080, 1920), (130, 516)])
def test_soak_fused_single_calls_under_uneven_load(wm, tc, shape):
    """the fused single-launch kernels, two alternating inputs, 300 synchronous embed + detect calls, while a second
    engine keeps the chip busy with batched sweeps on its own streams (the fused launch's workgroups then do not all start
    together: arrival order and timing of every hand-off vary from call to call).  Bitwise-equal results per input, no
    launch may time out."""
    torch = tc
    L = wm.lib()
    R, Cc = shape
    W = synth_watermark(R, Cc)
    xs = [torch.from_numpy(synth_frame(R, Cc, frame=f)).cuda() for f in range(2)]
    eng = wm.Watermark(R, Cc, W, 3, 40.0)
    assert eng.fused_info()[0]
    # background load: batched embed + detect on another engine (smaller frames, 2 slots)
    Rb, Cb, Fb = 540, 960, 8
    bg = wm.Watermark(Rb, Cb, synth_watermark(Rb, Cb), 3, 40.0, nslots=2, max_frames=Fb)
    xb = torch.from_numpy(np.stack([synth_frame(Rb, Cb, frame=10 + f) for f in range(Fb)])).cuda()
    yb = [torch.empty_like(xb) for _ in range(2)]
    stop = threading.Event()

    def background():
        pxb, pyb = wm.plane_of(xb), [wm.plane_of(y) for y in yb]
        while not stop.is_set():
            for s in range(2):
                bg.embed_async(pxb, pxb, pyb[s], wm.MASK_TYPE.ME, s)
                bg.detect_async(pyb[s], wm.MASK_TYPE.ME, s)
            for s in range(2):
                bg.sync(s)
    th = threading.Thread(target=background)
    th.start()
    try:
        y = torch.empty_like(xs[0])
        py = wm.plane_of(y)
        a, c = (C.c_float * 1)(), (C.c_float * 1)()
        refs = {}
        for it in range(300):
            k = it & 1
            mask = (it >> 1) & 1  # ME, ME, NVF, NVF, ...
            px = wm.plane_of(xs[k])
            assert L.wm_embed(eng._ctx, mask, C.byref(px), C.byref(px), C.byref(py), a, None, wm.WM_SLOT_SYNC) == 0
            assert L.wm_detect(eng._ctx, mask, C.byref(py), c, None, wm.WM_SLOT_SYNC) == 0
            cur = (a[0], c[0], int(y.view(torch.uint8).to(torch.int64).sum()) if it < 8 or it % 25 == 0 else None)
            ref = refs.setdefault((k, mask), cur)
            assert cur[0] == ref[0] and cur[1] == ref[1], (it, cur, ref)
            if cur[2] is not None and ref[2] is not None:
                assert cur[2] == ref[2], (it, "output checksum")
    finally:
        stop.set()
        th.join()
    assert eng.fused_info()[3] == 0, "a fused launch timed out under load"
    eng.close()
    bg.close()
