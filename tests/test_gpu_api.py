"""GPU tests of the C-ABI plumbing around the kernels: host-staged planes (pinned and pageable), caller-provided
streams, slot bookkeeping, argument errors at the boundary, RGB / batched RGB bases."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
from synth import synth_frame, synth_watermark

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tc():
    import torch
    assert torch.cuda.is_available()
    return torch


def host_plane(wm, arr, mem=None, pitch=None):
    mem = wm.WM_MEM_HOST if mem is None else mem
    dt = wm.WM_F32 if arr.dtype == np.float32 else wm.WM_U8
    rows, cols = arr.shape[-2], arr.shape[-1]
    ch = arr.shape[0] if arr.ndim == 3 else 1
    return wm.wm_plane(arr.ctypes.data, rows, cols, ch, dt, mem, 1, pitch or arr.strides[-2] // arr.itemsize,
                       arr.strides[0] // arr.itemsize if arr.ndim == 3 else 0, 0)


@pytest.mark.parametrize("pinned", [True, False])
@pytest.mark.parametrize("dtype", [np.uint8, np.float32])
def test_host_staged_planes(wm, tc, pinned, dtype):
    """WM_MEM_HOST planes: the library stages through its own device buffers with 2-D async copies (de-pitching like
    main.cpp:348-353) on the slot's stream; pinned memory comes from wm_host_alloc (main.cpp:273-275)"""
    L = wm.lib()
    R, Cc, pitch = 120, 300, 320
    W = synth_watermark(R, Cc)
    x = synth_frame(R, Cc, frame=4, dtype=dtype)
    eng = wm.Watermark(R, Cc, W, 3, 40.0)
    nbytes = R * pitch * x.itemsize
    if pinned:
        pin = L.wm_host_alloc(nbytes)
        pout = L.wm_host_alloc(nbytes)
        assert pin and pout
        hin = np.ctypeslib.as_array(C.cast(pin, C.POINTER(C.c_uint8)), shape=(nbytes,)).view(dtype).reshape(R, pitch)
        hout = np.ctypeslib.as_array(C.cast(pout, C.POINTER(C.c_uint8)), shape=(nbytes,)).view(dtype).reshape(R, pitch)
    else:
        hin = np.zeros((R, pitch), dtype)
        hout = np.zeros((R, pitch), dtype)
    hin[:] = 7
    hout[:] = 9
    hin[:, :Cc] = x
    p_in = host_plane(wm, hin[:, :Cc], pitch=pitch)
    p_out = host_plane(wm, hout[:, :Cc], pitch=pitch)
    a = (C.c_float * 1)()
    st = (C.c_int * 1)()
    corr = (C.c_float * 1)()
    rc = L.wm_embed(eng._ctx, 0, C.byref(p_in), C.byref(p_in), C.byref(p_out), a, st, 1)
    assert rc == 0
    rc = L.wm_detect(eng._ctx, 0, C.byref(p_out), corr, None, 1)   # same slot => ordered after the embed's D2H
    assert rc == 0
    assert L.wm_sync(eng._ctx, 1) == 0
    if dtype == np.uint8:
        so, yo, ao = O.embed_u8(x, W)
        d = np.abs(hout[:, :Cc].astype(int) - yo.astype(int))
        assert d.max() <= 1 and (d != 0).mean() <= 1e-3
        assert corr[0] == pytest.approx(O.detect_u8(np.ascontiguousarray(hout[:, :Cc]), W)[1], abs=1e-5)
    else:
        so, yo, ao = O.embed(x, x, W)
        np.testing.assert_allclose(hout[:, :Cc], yo, rtol=0, atol=1e-3)
        assert corr[0] == pytest.approx(O.detect(yo, W)[1], abs=1e-5)
    assert a[0] == pytest.approx(ao, rel=1e-4) and st[0] == 0
    assert (hout[:, Cc:] == 9).all() and (hin[:, Cc:] == 7).all()   # padding columns untouched
    if pinned:
        eng.close()
        L.wm_host_free(pin)
        L.wm_host_free(pout)


def test_caller_stream(wm, tc):
    """wm_set_stream: a slot runs on the caller's HIP stream (here a torch stream), ordered with the caller's own work"""
    torch = tc
    R, Cc = 128, 256
    W = synth_watermark(R, Cc)
    x = synth_frame(R, Cc)
    eng = wm.Watermark(R, Cc, W, 3, 40.0)
    L = wm.lib()
    s = torch.cuda.Stream()
    assert L.wm_set_stream(eng._ctx, 0, C.c_void_p(s.cuda_stream)) == 0
    assert L.wm_get_stream(eng._ctx, 0) == s.cuda_stream
    with torch.cuda.stream(s):
        xd = torch.from_numpy(x).cuda(non_blocking=True) * 1.0   # produced on the same stream, no host sync in between
        out = torch.empty_like(xd)
        a = (C.c_float * 1)()
        eng.embed_async(xd, xd, out, wm.MASK_TYPE.NVF, 0, a_out=a)
        doubled = out * 2.0                                      # consumer on the same stream
    assert eng.sync(0) == 0
    s.synchronize()
    so, yo, ao = O.embed(x, x, W, mask=O.MASK_NVF)
    assert a[0] == pytest.approx(ao, rel=1e-4)
    np.testing.assert_allclose(doubled.cpu().numpy(), 2 * yo, rtol=0, atol=2e-3)
    assert L.wm_set_stream(eng._ctx, 0, None) == 0


def test_argument_errors(wm, tc):
    torch = tc
    L = wm.lib()
    R, Cc = 64, 128
    eng = wm.Watermark(R, Cc, synth_watermark(R, Cc), 3, 40.0, nslots=2, max_frames=2)
    x = torch.zeros((R, Cc), device="cuda")
    small = torch.zeros((R, Cc - 1), device="cuda")
    with pytest.raises(RuntimeError, match="engine was initialised for"):
        eng.makeWatermark(small, small, wm.MASK_TYPE.ME)
    with pytest.raises(RuntimeError, match="bad slot"):
        eng.embed_async(x, x, x.clone(), wm.MASK_TYPE.ME, 5)
    big = torch.zeros((3, R, Cc), device="cuda")
    with pytest.raises(RuntimeError, match="max_frames"):
        eng.detectWatermark(big, wm.MASK_TYPE.ME)
    with pytest.raises(RuntimeError, match="same dtype"):
        eng.embed_async(x, x.to(torch.uint8), x.to(torch.uint8), wm.MASK_TYPE.ME, 0)
    with pytest.raises(RuntimeError):
        eng.makeWatermark(x.double(), x.double(), wm.MASK_TYPE.ME)
    assert L.wm_embed(eng._ctx, 7, None, None, None, None, None, 0) == wm.WM_ERR_BAD_ARG
    # RGB batch whose frames overlap (frame_stride covers one channel plane only): rejected, not raced
    rgb = torch.zeros((2, 3, R, Cc), device="cuda")
    pin = wm.plane_of(torch.zeros((2, R, Cc), device="cuda"), 1)
    pb = wm.plane_of(rgb, 3)
    pb.frame_stride = R * Cc
    with pytest.raises(RuntimeError, match="frame_stride too small"):
        eng.embed_async(pin, pb, pb, wm.MASK_TYPE.NVF, 0)
    assert L.wm_sync(eng._ctx, 9) == wm.WM_ERR_BAD_ARG
    assert b"bad slot" in L.wm_last_error(eng._ctx)
    # the engine is still usable
    xf = synth_frame(R, Cc)
    y, a = eng.makeWatermark(torch.from_numpy(xf).cuda(), torch.from_numpy(xf).cuda(), wm.MASK_TYPE.ME)
    assert a == pytest.approx(O.embed(xf, xf, synth_watermark(R, Cc))[2], rel=1e-4)


@pytest.mark.parametrize("dtype", ["f32", "u8"])
def test_rgb_bases_batched(wm, tc, dtype):
    """planar RGB base, batch of frames, both dtypes: a*u is added to all three channels (main.cpp:169-190)"""
    torch = tc
    R, Cc, F = 70, 260, 3
    W = synth_watermark(R, Cc)
    eng = wm.Watermark(R, Cc, W, 3, 40.0, nslots=1, max_frames=F)
    npdt = np.float32 if dtype == "f32" else np.uint8
    gray = np.stack([synth_frame(R, Cc, frame=f, dtype=npdt) for f in range(F)])
    rgb = np.stack([np.stack([synth_frame(R, Cc, frame=10 + 3 * f + ch, dtype=npdt) for ch in range(3)]) for f in range(F)])
    for mt in (wm.MASK_TYPE.ME, wm.MASK_TYPE.NVF):
        y, a = eng.makeWatermark(torch.from_numpy(gray).cuda(), torch.from_numpy(rgb).cuda(), mt)
        assert tuple(y.shape) == (F, 3, R, Cc)
        for f in range(F):
            so, yo, ao = O.embed(gray[f].astype(np.float32), rgb[f].astype(np.float32), W, mask=int(mt))
            assert a[f] == pytest.approx(ao, rel=1e-4)
            if dtype == "f32":
                np.testing.assert_allclose(y[f].cpu().numpy(), yo, rtol=0, atol=1e-3)
            else:
                d = np.abs(y[f].cpu().numpy().astype(int) - yo.astype(np.uint8).astype(int))
                assert d.max() <= 1 and (d != 0).mean() <= 1e-3


def test_many_ops_per_slot_before_sync(wm, tc):
    """results are delivered in order for every op queued on a slot since the last sync"""
    torch = tc
    R, Cc = 64, 256
    W = synth_watermark(R, Cc)
    eng = wm.Watermark(R, Cc, W, 3, 40.0)
    xs = [torch.from_numpy(synth_frame(R, Cc, frame=f)).cuda() for f in range(6)]
    outs = [torch.empty_like(x) for x in xs]
    a = [(C.c_float * 1)() for _ in xs]
    corr = [(C.c_float * 1)() for _ in xs]
    torch.cuda.synchronize()
    for f, x in enumerate(xs):
        eng.embed_async(x, x, outs[f], wm.MASK_TYPE.ME, 0, a_out=a[f])
        eng.detect_async(outs[f], wm.MASK_TYPE.ME, 0, corr_out=corr[f])
    assert eng.sync(0) == 0
    for f, x in enumerate(xs):
        xh = x.cpu().numpy()
        so, yo, ao = O.embed(xh, xh, W)
        assert a[f][0] == pytest.approx(ao, rel=1e-4)
        assert corr[f][0] == pytest.approx(O.detect(yo, W)[1], abs=1e-5)


def test_fold_tails_across_ops_shapes_and_slots(wm, tc):
    """The fold steps (solve, strength, correlation) are run by the block that finishes a frame's sweep last, found
    with a per-frame ticket that must be back at zero for the next op on the slot.  Ops of varying batch sizes, masks
    and kinds are interleaved on two slots of one engine -- on a shape whose sweeps take two launches (aligned strips
    + a ragged strip) and whose segments make several blocks per strip -- and every result must equal the result of
    the same frame processed alone on a fresh engine, bit for bit."""
    torch = tc
    R, Cc = 300, 700
    W = synth_watermark(R, Cc)
    frames = np.stack([synth_frame(R, Cc, frame=f) for f in range(6)])
    frames[4] = 77.0  # unsolvable frame in the middle of a batch
    ref_eng = wm.Watermark(R, Cc, W, 3, 40.0)
    ref_eng.set_rows_per_segment(16)  # same partition of the partial sums as below (sums are f32 per thread)
    ref = {}
    for f in range(6):
        for mk in (wm.MASK_TYPE.ME, wm.MASK_TYPE.NVF):
            x = torch.from_numpy(frames[f]).cuda()
            y, a = ref_eng.makeWatermark(x, x, mk)
            ref[(f, int(mk))] = (y.cpu().numpy(), a, ref_eng.detectWatermark(y, mk))
    ref_eng.close()
    assert ref[(4, 0)][1] is None and ref[(4, 0)][2] == 0.0  # ME on the constant frame: passthrough

    eng = wm.Watermark(R, Cc, W, 3, 40.0, nslots=2, max_frames=6)
    eng.set_rows_per_segment(16)  # 19 segments -> 5 blocks per strip, 3 strips: 15 march blocks + border blocks per frame
    dev = torch.from_numpy(frames).cuda()
    rng = np.random.default_rng(5)
    queued = []
    for it in range(24):
        F = int(rng.integers(1, 7))
        first = int(rng.integers(0, 7 - F))
        mk = wm.MASK_TYPE.ME if it % 3 else wm.MASK_TYPE.NVF
        slot = it % 2
        xb = dev[first:first + F]
        yb = torch.empty_like(xb)
        a = (C.c_float * F)(*([float("nan")] * F))
        st = (C.c_int * F)()
        corr = (C.c_float * F)()
        torch.cuda.synchronize()
        eng.embed_async(xb, xb, yb, mk, slot, a_out=a, status_out=st)
        eng.detect_async(yb, mk, slot, corr_out=corr)
        queued.append((first, F, mk, yb, a, st, corr))
        if it % 4 == 3:
            eng.sync(0)
            eng.sync(1)
            eng.computeMask(xb[0], mk)  # a mask-only op in between (the stats sweep and its tail on slot 0)
            for first_q, Fq, mkq, ybq, aq, stq, corrq in queued:
                for k in range(Fq):
                    yr, ar, cr = ref[(first_q + k, int(mkq))]
                    np.testing.assert_array_equal(ybq[k].cpu().numpy(), yr)
                    if ar is None:
                        assert stq[k] != 0 and np.isnan(aq[k])
                    else:
                        assert stq[k] == 0 and aq[k] == ar
                    # (the two paths group the per-thread f32 partial sums differently: the score may differ in its last bit)
                    assert abs(corrq[k] - cr) <= 1.2e-7
            queued = []
    eng.close()


@pytest.mark.parametrize("frames", [1, 3])
def test_slot_output_plane(wm, tc, frames):
    """WM_MEM_SLOT_OUT: wm_detect on the device copy of what the last wm_embed of the slot wrote -- a host-staged frame is
    detected without a second trip over the host link; the score equals the one of the downloaded output"""
    torch = tc
    L = wm.lib()
    R, Cc = 120, 516
    n = R * Cc
    W = synth_watermark(R, Cc)
    eng = wm.Watermark(R, Cc, W, 3, 40.0, nslots=2, max_frames=frames)
    xs = np.stack([synth_frame(R, Cc, frame=f, dtype=np.uint8) for f in range(frames)])
    ys = np.empty_like(xs)

    def hp(a):
        return wm.wm_plane(a.ctypes.data, R, Cc, 1, wm.WM_U8, wm.WM_MEM_HOST, frames, Cc, 0, n)
    slot_plane = wm.wm_plane(None, R, Cc, 1, wm.WM_U8, wm.WM_MEM_SLOT_OUT, frames, Cc, 0, n)
    corr_slot, corr_host = (C.c_float * frames)(), (C.c_float * frames)()
    # nothing embedded on slot 1 yet
    assert L.wm_detect(eng._ctx, 0, C.byref(slot_plane), corr_slot, None, 1) == wm.WM_ERR_BAD_ARG
    pin, pout = hp(xs), hp(ys)
    a = (C.c_float * frames)()
    assert L.wm_embed(eng._ctx, 0, C.byref(pin), C.byref(pin), C.byref(pout), a, None, 1) == 0
    assert L.wm_detect(eng._ctx, 0, C.byref(slot_plane), corr_slot, None, 1) == 0
    assert L.wm_sync(eng._ctx, 1) == 0
    assert L.wm_detect(eng._ctx, 0, C.byref(pout), corr_host, None, 1) == 0
    assert L.wm_sync(eng._ctx, 1) == 0
    for f in range(frames):
        assert corr_slot[f] == corr_host[f]
        assert corr_slot[f] == pytest.approx(O.detect_u8(ys[f], W)[1], abs=1e-5)
    # a plane that does not match the slot's last embed is refused
    bad = wm.wm_plane(None, R, Cc, 1, wm.WM_F32, wm.WM_MEM_SLOT_OUT, frames, Cc, 0, n)
    assert L.wm_detect(eng._ctx, 0, C.byref(bad), corr_slot, None, 1) == wm.WM_ERR_BAD_ARG
    eng.close()


@pytest.mark.parametrize("frames", [1, 2])
def test_slot_output_plane_is_an_input_only_and_in_place_through_the_slot(wm, tc, frames):
    """WM_MEM_SLOT_OUT has no address of its own: as `base` or `out` it is refused (it used to reach the kernels as a null
    pointer).  As the INPUT of a second embed whose `out` is the buffer the slot's last output lives in, the call is in place
    on the resolved addresses: the stencil must read the original pixels (snapshot on the sweeps), so the result equals the
    same embed run out of place."""
    torch = tc
    L = wm.lib()
    R, Cc = 90, 516
    n = R * Cc
    W = synth_watermark(R, Cc)
    eng = wm.Watermark(R, Cc, W, 3, 40.0, nslots=1, max_frames=frames)
    eng.set_fused(False)
    xs = torch.from_numpy(np.stack([synth_frame(R, Cc, frame=f) for f in range(frames)])).cuda()
    y1 = torch.empty_like(xs)
    px, py1 = wm.plane_of(xs), wm.plane_of(y1)
    slot_plane = wm.wm_plane(None, R, Cc, 1, wm.WM_F32, wm.WM_MEM_SLOT_OUT, frames, Cc, 0, n)
    a = (C.c_float * frames)()
    assert L.wm_embed(eng._ctx, 0, C.byref(px), C.byref(px), C.byref(py1), a, None, 0) == 0
    assert L.wm_sync(eng._ctx, 0) == 0
    for what, args in (("base", (px, slot_plane, py1)), ("out", (px, px, slot_plane))):
        rc = L.wm_embed(eng._ctx, 0, C.byref(args[0]), C.byref(args[1]), C.byref(args[2]), a, None, 0)
        assert rc == wm.WM_ERR_BAD_ARG, what
        assert b"WM_MEM_SLOT_OUT" in L.wm_last_error(eng._ctx)
    # reference: second embed of y1 out of place
    y2 = torch.empty_like(xs)
    py2 = wm.plane_of(y2)
    a2 = (C.c_float * frames)()
    assert L.wm_embed(eng._ctx, 1, C.byref(py1), C.byref(py1), C.byref(py2), a2, None, 0) == 0
    assert L.wm_sync(eng._ctx, 0) == 0
    y1_copy = y1.clone()
    # re-create the slot's "last output = y1" state, then embed in = SLOT_OUT, base = y1, out = y1 (in place through the slot)
    assert L.wm_embed(eng._ctx, 0, C.byref(px), C.byref(px), C.byref(py1), a, None, 0) == 0
    assert L.wm_sync(eng._ctx, 0) == 0
    assert torch.equal(y1, y1_copy)
    a3 = (C.c_float * frames)()
    assert L.wm_embed(eng._ctx, 1, C.byref(slot_plane), C.byref(py1), C.byref(py1), a3, None, 0) == 0
    assert L.wm_sync(eng._ctx, 0) == 0
    assert list(a3) == list(a2)
    assert torch.equal(y1, y2), "in-place embed through WM_MEM_SLOT_OUT read pixels it had already overwritten"
    eng.close()
