"""The fused single-frame kernels (wm_k_fused.hip: one launch per makeWatermark / detectWatermark call, tiles resident in
LDS) against the CPU oracle and against the batched sweeps on the same inputs, over the tile geometries the launch
can produce: partial last row band, tile heights that are not a multiple of the rows per wavefront, shifted last strip,
4 and 8 rows per wavefront, images of a few rows; both element types, both masks, RGB bases, in-place frames, host planes,
the unsolvable passthrough, run-to-run determinism, and two host threads calling at once."""
import ctypes as C
import threading

import numpy as np
import pytest

import oracle_lib as O
from synth import synth_frame, synth_watermark

pytestmark = pytest.mark.gpu

TOL_A, TOL_CORR, TOL_Y = 1e-4, 1e-5, 1e-3

# (rows, cols): all have cols % 4 == 0 and cols >= 256 (what the fused path takes)
SHAPES = [(4, 256), (5, 260), (9, 512), (37, 256), (64, 300), (130, 516), (257, 764), (300, 1028), (1000, 1280), (1080, 1920)]
# ... and the size the single-call headline is quoted on (255 workgroups of 256 x 128-row tiles: the largest fused geometry)
SHAPES_4K = SHAPES + [(2160, 3840)]


@pytest.fixture(scope="module")
def tc():
    import torch
    assert torch.cuda.is_available()
    return torch


def dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def engines(wm, R, Cc, W, p=3):
    ef = wm.Watermark(R, Cc, W, p, 40.0)
    es = wm.Watermark(R, Cc, W, p, 40.0)
    ef.set_fused(True)
    es.set_fused(False)
    assert ef.fused_info()[0], f"{R}x{Cc} should take the fused kernels"
    assert not es.fused_info()[0]
    return ef, es


@pytest.mark.parametrize("shape", SHAPES_4K)
@pytest.mark.parametrize("mask", ["ME", "NVF"])
def test_fused_f32_vs_oracle_and_sweeps(wm, tc, shape, mask):
    torch = tc
    R, Cc = shape
    mk, omk = (wm.MASK_TYPE.ME, O.MASK_ME) if mask == "ME" else (wm.MASK_TYPE.NVF, O.MASK_NVF)
    x = synth_frame(R, Cc, frame=1)
    W = synth_watermark(R, Cc)
    ef, es = engines(wm, R, Cc, W)
    ef.prof_enable(True)
    xd = dev(torch, x)
    yf, af = ef.makeWatermark(xd, xd, mk)
    ys, as_ = es.makeWatermark(xd, xd, mk)
    so, yo, ao = O.embed(x, x, W, mask=omk)
    assert so == 0
    assert af == pytest.approx(ao, rel=TOL_A) and af == pytest.approx(as_, rel=1e-6)
    np.testing.assert_allclose(yf.cpu().numpy(), yo, rtol=0, atol=TOL_Y)
    np.testing.assert_allclose(yf.cpu().numpy(), ys.cpu().numpy(), rtol=0, atol=2e-4)
    cf = ef.detectWatermark(dev(torch, yo), mk)
    cs = es.detectWatermark(dev(torch, yo), mk)
    co = O.detect(yo, W, mask=omk)[1]
    assert cf == pytest.approx(co, abs=TOL_CORR) and cf == pytest.approx(cs, abs=2e-6)
    # an unmarked image scores near zero on both paths
    assert ef.detectWatermark(xd, mk) == pytest.approx(O.detect(x, W, mask=omk)[1], abs=TOL_CORR)
    rep = ef.prof_report()
    assert "k_fused_embed" in rep and "k_fused_detect" in rep and "k_gram" not in rep, rep
    assert ef.fused_info()[3] == 0, "a fused launch timed out and fell back"
    ef.close(); es.close()


@pytest.mark.parametrize("shape", SHAPES_4K)
@pytest.mark.parametrize("dtype", ["f32", "u8"])
def test_fused_gram_sums_exact(wm, tc, shape, dtype, monkeypatch):
    """the 44 Gram sums the fused kernel folds (lag sums of the tiles + border chunks, recursive-halving wave reductions,
    term-major records, row-wise fold) against the oracle's f64 Gram: exact products summed in f64 on both sides, so they
    agree to reduction-order rounding"""
    torch = tc
    monkeypatch.setenv("WM_FUSED_STAMPS", "1")
    R, Cc = shape
    x = synth_frame(R, Cc, frame=3, dtype=np.uint8 if dtype == "u8" else np.float32)
    eng = wm.Watermark(R, Cc, synth_watermark(R, Cc), 3, 40.0)
    assert eng.fused_info()[0]
    eng.detectWatermark(dev(torch, x), wm.MASK_TYPE.ME)
    buf = (C.c_double * 44)()
    assert wm.lib().wm_fused_gram(eng._ctx, buf) == 44
    tot = np.array(buf[:])
    Ro, ro = O.gram(x.astype(np.float32))
    ref = np.concatenate([np.array([Ro[i, j] for i in range(8) for j in range(i, 8)]), ro])
    np.testing.assert_allclose(tot, ref, rtol=1e-13, atol=0)
    eng.close()


@pytest.mark.parametrize("shape", [(6, 256), (98, 300), (135, 516), (270, 1024), (720, 1280), (2160, 3840)])
@pytest.mark.parametrize("mask", ["ME", "NVF"])
def test_fused_u8_frames(wm, tc, shape, mask):
    """video Y planes (u8 in, u8 out by truncation, main.cpp:355-357), embedded in place like the reference's loop"""
    torch = tc
    R, Cc = shape
    mk, omk = (wm.MASK_TYPE.ME, O.MASK_ME) if mask == "ME" else (wm.MASK_TYPE.NVF, O.MASK_NVF)
    x = synth_frame(R, Cc, frame=4, dtype=np.uint8)
    W = synth_watermark(R, Cc)
    ef, es = engines(wm, R, Cc, W)
    so, yo, ao = O.embed_u8(x, W, mask=omk)
    xd = dev(torch, x)
    yf, af = ef.makeWatermark(xd, xd, mk)
    d = np.abs(yf.cpu().numpy().astype(int) - yo.astype(int))
    assert d.max() <= 1 and (d != 0).mean() <= 2e-3
    assert af == pytest.approx(ao, rel=TOL_A)
    # in place: input, base and output are the same frame (main.cpp:356,380); no snapshot is taken on this path
    frame = xd.clone()
    y2, a2 = ef.makeWatermark(frame, frame, mk, out=frame)
    assert torch.equal(frame, yf) and a2 == af
    ys, _ = es.makeWatermark(xd, xd, mk)
    ds = np.abs(yf.cpu().numpy().astype(int) - ys.cpu().numpy().astype(int))
    assert ds.max() <= 1 and (ds != 0).mean() <= 1e-3
    cf = ef.detectWatermark(dev(torch, yo), mk)
    assert cf == pytest.approx(O.detect_u8(yo, W, mask=omk)[1], abs=TOL_CORR)
    assert cf == pytest.approx(es.detectWatermark(dev(torch, yo), mk), abs=2e-6)
    assert ef.fused_info()[3] == 0
    ef.close(); es.close()


@pytest.mark.parametrize("dtype", ["f32", "u8"])
def test_fused_rgb_base_and_separate_grey_base(wm, tc, dtype):
    """a * u added to all three channels of a planar RGB base (main.cpp:169-190), and a grey base that is not the input"""
    torch = tc
    R, Cc = 190, 772
    u8 = dtype == "u8"
    x = synth_frame(R, Cc, frame=2, dtype=np.uint8 if u8 else np.float32)
    rgb = np.stack([synth_frame(R, Cc, frame=10 + k, dtype=np.uint8 if u8 else np.float32) for k in range(3)])
    W = synth_watermark(R, Cc)
    ef, es = engines(wm, R, Cc, W)
    for mk in (wm.MASK_TYPE.ME, wm.MASK_TYPE.NVF):
        yf, af = ef.makeWatermark(dev(torch, x), dev(torch, rgb), mk)
        ys, as_ = es.makeWatermark(dev(torch, x), dev(torch, rgb), mk)
        assert af == pytest.approx(as_, rel=1e-6)
        if u8:
            d = np.abs(yf.cpu().numpy().astype(int) - ys.cpu().numpy().astype(int))
            assert d.max() <= 1 and (d != 0).mean() <= 1e-3
        else:
            so, yo, ao = O.embed(x, rgb, W, mask=int(mk))
            assert af == pytest.approx(ao, rel=TOL_A)
            np.testing.assert_allclose(yf.cpu().numpy(), yo, rtol=0, atol=TOL_Y)
        g = rgb[1]
        yg, ag = ef.makeWatermark(dev(torch, x), dev(torch, g), mk)
        ysg, _ = es.makeWatermark(dev(torch, x), dev(torch, g), mk)
        assert ag == af
        if u8:
            assert np.abs(yg.cpu().numpy().astype(int) - ysg.cpu().numpy().astype(int)).max() <= 1
        else:
            np.testing.assert_allclose(yg.cpu().numpy(), ysg.cpu().numpy(), rtol=0, atol=2e-4)
    assert ef.fused_info()[3] == 0
    ef.close(); es.close()


def test_fused_unsolvable_passthrough_and_pitched_planes(wm, tc):
    torch = tc
    R, Cc = 100, 512
    W = synth_watermark(R, Cc)
    ef, es = engines(wm, R, Cc, W)
    flat = torch.full((R, Cc), 77.0, device="cuda")
    y, a = ef.makeWatermark(flat, flat, wm.MASK_TYPE.ME)
    assert a is None and torch.equal(y, flat)                      # Watermark.cpp:164-165
    assert ef.detectWatermark(flat, wm.MASK_TYPE.ME) == 0.0        # Watermark.cpp:246-247
    base = torch.rand((3, R, Cc), device="cuda") * 255
    y, a = ef.makeWatermark(flat, base, wm.MASK_TYPE.ME)
    assert a is None and torch.equal(y, base)
    # pitched (but vector-aligned) planes: views into wider buffers
    x = synth_frame(R, Cc, frame=6)
    big = torch.zeros((R, Cc + 64), device="cuda")
    big[:, 32:32 + Cc] = dev(torch, x)
    view = big[:, 32:32 + Cc]
    outbig = torch.zeros((R, Cc + 128), device="cuda")
    oview = outbig[:, 64:64 + Cc]
    y, a = ef.makeWatermark(view, view, wm.MASK_TYPE.ME, out=oview)
    so, yo, ao = O.embed(x, x, W)
    assert a == pytest.approx(ao, rel=TOL_A)
    np.testing.assert_allclose(oview.cpu().numpy(), yo, rtol=0, atol=TOL_Y)
    assert float(outbig[:, :64].abs().max()) == 0.0 and float(outbig[:, 64 + Cc:].abs().max()) == 0.0
    assert ef.detectWatermark(oview, wm.MASK_TYPE.ME) == pytest.approx(O.detect(oview.cpu().numpy(), W)[1], abs=TOL_CORR)
    # an unaligned view falls back to the sweeps and still answers
    odd = big[:, 33:33 + Cc]
    y3, a3 = ef.makeWatermark(odd, odd, wm.MASK_TYPE.ME)
    so, yo3, ao3 = O.embed(odd.cpu().numpy(), odd.cpu().numpy(), W)
    assert a3 == pytest.approx(ao3, rel=TOL_A)
    assert ef.fused_info()[3] == 0
    ef.close(); es.close()


@pytest.mark.parametrize("shape", [(64, 257), (64, 258), (70, 259), (98, 301), (135, 518), (139, 769), (60, 770), (270, 1023), (130, 1025), (1078, 1918), (2160, 3838)])
@pytest.mark.parametrize("mask", ["ME", "NVF"])
def test_fused_widths_that_are_not_multiples_of_4(wm, tc, shape, mask, monkeypatch):
    """f32 planes of any width >= 256 (the reference's 4k_non_divisible sample is 3838 wide): the image's last strip ends at the
    last column wherever that is -- 16-byte accesses at 4-byte aligned addresses, one lane of that strip split between owned
    and duplicate pixels.  Against the oracle and the sweeps: y, strength, the 44 Gram sums, the scores, in place, the one-call
    pair.  u8 planes of such a width keep the sweeps (a lane's 4 pixels are one dword)."""
    torch = tc
    monkeypatch.setenv("WM_FUSED_STAMPS", "1")   # (keeps the folded Gram sums readable: wm_fused_gram)
    R, Cc = shape
    mk, omk = (wm.MASK_TYPE.ME, O.MASK_ME) if mask == "ME" else (wm.MASK_TYPE.NVF, O.MASK_NVF)
    x = synth_frame(R, Cc, frame=6)
    W = synth_watermark(R, Cc)
    ef, es = engines(wm, R, Cc, W)
    ef.prof_enable(True)
    xd = dev(torch, x)
    yf, af = ef.makeWatermark(xd, xd, mk)
    ys, as_ = es.makeWatermark(xd, xd, mk)
    so, yo, ao = O.embed(x, x, W, mask=omk)
    assert so == 0 and af == pytest.approx(ao, rel=TOL_A) and af == pytest.approx(as_, rel=1e-6)
    np.testing.assert_allclose(yf.cpu().numpy(), yo, rtol=0, atol=TOL_Y)
    np.testing.assert_allclose(yf.cpu().numpy(), ys.cpu().numpy(), rtol=0, atol=2e-4)
    cf = ef.detectWatermark(dev(torch, yo), mk)
    assert cf == pytest.approx(O.detect(yo, W, mask=omk)[1], abs=TOL_CORR) and cf == pytest.approx(es.detectWatermark(dev(torch, yo), mk), abs=2e-6)
    buf = (C.c_double * 44)()
    assert wm.lib().wm_fused_gram(ef._ctx, buf) == 44
    Ro, ro = O.gram(yo.astype(np.float32))
    np.testing.assert_allclose(np.array(buf[:]), np.concatenate([np.array([Ro[i, j] for i in range(8) for j in range(i, 8)]), ro]), rtol=1e-13, atol=0)
    assert ef.detectWatermark(xd, mk) == pytest.approx(O.detect(x, W, mask=omk)[1], abs=TOL_CORR)
    # in place (main.cpp:356,380) and the pair as one call
    frame = xd.clone()
    y2, a2 = ef.makeWatermark(frame, frame, mk, out=frame)
    assert torch.equal(frame, yf) and a2 == af
    y3, a3, c3 = ef.makeAndDetect(xd, xd, mk)
    assert torch.equal(y3, yf) and a3 == af and c3 == ef.detectWatermark(yf, mk)
    rep = ef.prof_report()
    assert "k_fused_embed" in rep and "k_fused_detect" in rep and "k_gram" not in rep, rep
    assert ef.fused_info()[3] == 0, "a fused launch timed out and fell back"
    if mask == "ME" and R <= 300:
        # an RGB base and a grey base that is not the input
        rgb = np.stack([synth_frame(R, Cc, frame=20 + k) for k in range(3)])
        yr, ar = ef.makeWatermark(xd, dev(torch, rgb), mk)
        sor, yor, aor = O.embed(x, rgb, W, mask=omk)
        assert ar == pytest.approx(aor, rel=TOL_A)
        np.testing.assert_allclose(yr.cpu().numpy(), yor, rtol=0, atol=TOL_Y)
        # u8 frames of this width: the sweeps, same answers as ever
        xu = synth_frame(R, Cc, frame=6, dtype=np.uint8)
        before = ef.prof_report()
        yu, au = ef.makeWatermark(dev(torch, xu), dev(torch, xu), mk)
        sou, you, aou = O.embed_u8(xu, W, mask=omk)
        d = np.abs(yu.cpu().numpy().astype(int) - you.astype(int))
        assert d.max() <= 1 and (d != 0).mean() <= 2e-3 and au == pytest.approx(aou, rel=TOL_A)
        after = ef.prof_report()
        assert after["k_fused_embed"] == before["k_fused_embed"] and "k_gram" in after, (before, after)
        # host planes of this width through the library's staging buffers: the same bits as the device planes
        xh = np.ascontiguousarray(x)
        yh = np.empty_like(xh)
        hp = lambda arr: wm.wm_plane(arr.ctypes.data, R, Cc, 1, wm.WM_F32, wm.WM_MEM_HOST, 1, Cc, 0, 0)
        av, cv = (C.c_float * 1)(), (C.c_float * 1)()
        pin, pout = hp(xh), hp(yh)
        assert wm.lib().wm_embed(ef._ctx, int(mk), C.byref(pin), C.byref(pin), C.byref(pout), av, None, wm.WM_SLOT_SYNC) == 0
        assert av[0] == af
        np.testing.assert_array_equal(yh, yf.cpu().numpy())
        assert wm.lib().wm_detect(ef._ctx, int(mk), C.byref(pout), cv, None, wm.WM_SLOT_SYNC) == 0
        assert cv[0] == ef.detectWatermark(yf, mk)
    ef.close(); es.close()


def test_fused_not_taken_when_shape_does_not_fit(wm, tc):
    """widths below 256, p != 3 and row bands take the sweeps"""
    for (R, Cc, p) in [(64, 255, 3), (64, 128, 3), (64, 512, 5)]:
        e = wm.Watermark(R, Cc, synth_watermark(R, Cc), p, 40.0)
        assert not e.fused_info()[0], (R, Cc, p)
        x = tc.from_numpy(synth_frame(R, Cc)).cuda()
        y, a = e.makeWatermark(x, x, wm.MASK_TYPE.NVF)
        assert a == pytest.approx(O.embed(synth_frame(R, Cc), synth_frame(R, Cc), synth_watermark(R, Cc), p=p, mask=O.MASK_NVF)[2], rel=TOL_A)
        e.close()


def test_fused_determinism_and_host_planes(wm, tc):
    torch = tc
    R, Cc = 540, 960
    x = synth_frame(R, Cc, frame=9)
    W = synth_watermark(R, Cc)
    ef, es = engines(wm, R, Cc, W)
    xd = dev(torch, x)
    y0, a0 = ef.makeWatermark(xd, xd, wm.MASK_TYPE.ME)
    c0 = ef.detectWatermark(y0, wm.MASK_TYPE.ME)
    for _ in range(20):
        y, a = ef.makeWatermark(xd, xd, wm.MASK_TYPE.ME)
        assert a == a0 and torch.equal(y, y0)
        assert ef.detectWatermark(y, wm.MASK_TYPE.ME) == c0
    # host planes through the library's staging buffers
    L = wm.lib()
    xh = np.ascontiguousarray(x)
    yh = np.empty_like(xh)
    def hp(a):
        return wm.wm_plane(a.ctypes.data, R, Cc, 1, wm.WM_F32, wm.WM_MEM_HOST, 1, Cc, 0, 0)
    av, cv = (C.c_float * 1)(), (C.c_float * 1)()
    pin, pout = hp(xh), hp(yh)
    assert L.wm_embed(ef._ctx, 0, C.byref(pin), C.byref(pin), C.byref(pout), av, None, wm.WM_SLOT_SYNC) == 0
    assert av[0] == a0
    np.testing.assert_array_equal(yh, y0.cpu().numpy())
    assert L.wm_detect(ef._ctx, 0, C.byref(pout), cv, None, wm.WM_SLOT_SYNC) == 0
    assert cv[0] == c0
    assert ef.fused_info()[3] == 0
    ef.close(); es.close()


def test_fused_calls_from_two_host_threads(wm, tc):
    """two engines on one device called from two host threads at once: fused launches are serialised per device, every
    result equals the single-threaded one and no launch times out"""
    torch = tc
    R, Cc = 400, 1024
    W = synth_watermark(R, Cc)
    xs = [dev(torch, synth_frame(R, Cc, frame=f)) for f in range(2)]
    engs = [wm.Watermark(R, Cc, W, 3, 40.0) for _ in range(2)]
    ref = []
    for e, x in zip(engs, xs):
        y, a = e.makeWatermark(x, x, wm.MASK_TYPE.ME)
        ref.append((y.clone(), a, e.detectWatermark(y, wm.MASK_TYPE.ME)))
    torch.cuda.synchronize()
    errs = []

    def work(k):
        try:
            L = wm.lib()
            e, x = engs[k], xs[k]
            y = torch.empty_like(x)
            px, py = wm.plane_of(x), wm.plane_of(y)
            a, c = (C.c_float * 1)(), (C.c_float * 1)()
            for _ in range(150):
                assert L.wm_embed(e._ctx, 0, C.byref(px), C.byref(px), C.byref(py), a, None, wm.WM_SLOT_SYNC) == 0
                assert L.wm_detect(e._ctx, 0, C.byref(py), c, None, wm.WM_SLOT_SYNC) == 0
                assert a[0] == ref[k][1] and c[0] == ref[k][2]
            assert torch.equal(y, ref[k][0])
        except Exception as ex:  # noqa: BLE001
            errs.append(ex)
    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    for e in engs:
        assert e.fused_info()[3] == 0
        e.close()


def test_fused_timeout_falls_back_to_the_sweeps(wm, tc, monkeypatch):
    """a hand-off that cannot complete (test hook WM_FUSED_DBG=4: workgroup 0 never arrives -- what a workgroup that is not
    resident looks like to the others): every spin is bounded (milliseconds), the launch ends without a result, the host
    notices the END of the launch (it no longer sits out its own 200 ms limit), re-runs the call on the batched sweeps and
    answers correctly; then the context BACKS OFF: the next calls go straight to the sweeps, a later call probes the fused
    path again, and the window doubles while the probes keep failing"""
    import time
    torch = tc
    monkeypatch.setenv("WM_FUSED_DBG", "4")
    R, Cc = 130, 516
    x = synth_frame(R, Cc, frame=2)
    W = synth_watermark(R, Cc)
    eng = wm.Watermark(R, Cc, W, 3, 40.0)
    assert eng.fused_info()[0]
    xd = dev(torch, x)
    so, yo, ao = O.embed(x, x, W, mask=O.MASK_ME)
    co = O.detect(yo, W, mask=O.MASK_ME)[1]
    yod = dev(torch, yo)
    times = []
    for k in range(30):
        t0 = time.perf_counter()
        if k % 2 == 0:
            y, a = eng.makeWatermark(xd, xd, wm.MASK_TYPE.ME)
            times.append(time.perf_counter() - t0)
            assert a == pytest.approx(ao, rel=TOL_A)
            np.testing.assert_allclose(y.cpu().numpy(), yo, rtol=0, atol=TOL_Y)
        else:
            c = eng.detectWatermark(yod, wm.MASK_TYPE.ME)
            times.append(time.perf_counter() - t0)
            assert c == pytest.approx(co, abs=TOL_CORR)
    # call 0 times out (fallback 1, the next 8 calls skip the fused path), call 9 probes again (fallback 2, 16 calls skipped),
    # call 26 probes again (fallback 3)
    assert eng.fused_info()[3] == 3, eng.fused_info()
    assert times[0] < 0.1, f"a timed-out fused call took {times[0] * 1e3:.1f} ms (the host must notice the end of the launch)"
    # calls inside a back-off window pay no time-out (medians: a single call may hit a host hiccup or a first-use code load)
    med = lambda v: sorted(v)[len(v) // 2]
    assert med(times[1:9]) < 0.5 * times[0] and med(times[10:26]) < 0.5 * times[9], times
    eng.close()
    # NVF: the statistics hand-off is the one that cannot complete
    eng = wm.Watermark(R, Cc, W, 3, 40.0)
    y, a = eng.makeWatermark(xd, xd, wm.MASK_TYPE.NVF)
    so, yn, an = O.embed(x, x, W, mask=O.MASK_NVF)
    assert a == pytest.approx(an, rel=TOL_A)
    np.testing.assert_allclose(y.cpu().numpy(), yn, rtol=0, atol=TOL_Y)
    assert eng.fused_info()[3] == 1
    eng.close()


def test_fused_completion_not_observed_never_reruns_an_in_place_embed(wm, tc, monkeypatch):
    """test hook WM_FUSED_DBG=8: workgroup 0 writes its part of y but its end-of-embed flag is never seen.  The folding
    workgroup then reports FUSED_INCOMPLETE instead of leaving the record untouched: an OUT-OF-PLACE call is re-run on the
    sweeps (input and base are intact) and answers correctly; an IN-PLACE call (the video contract, main.cpp:356,380) must
    fail loudly -- re-running it would watermark an already watermarked frame."""
    torch = tc
    monkeypatch.setenv("WM_FUSED_DBG", "8")
    R, Cc = 130, 516
    x = synth_frame(R, Cc, frame=2)
    W = synth_watermark(R, Cc)
    eng = wm.Watermark(R, Cc, W, 3, 40.0)
    xd = dev(torch, x)
    so, yo, ao = O.embed(x, x, W, mask=O.MASK_ME)
    y, a = eng.makeWatermark(xd, xd, wm.MASK_TYPE.ME)       # out of place
    assert a == pytest.approx(ao, rel=TOL_A)
    np.testing.assert_allclose(y.cpu().numpy(), yo, rtol=0, atol=TOL_Y)
    assert eng.fused_info()[3] == 1
    eng.close()
    eng = wm.Watermark(R, Cc, W, 3, 40.0)
    frame = xd.clone()
    with pytest.raises(RuntimeError, match="completion of the output stores was not observed"):
        eng.makeWatermark(frame, frame, wm.MASK_TYPE.ME, out=frame)
    assert eng.fused_info()[3] == 1
    # what the failed call left behind is the ONCE-watermarked frame (every workgroup did store), never a twice-watermarked one
    np.testing.assert_allclose(frame.cpu().numpy(), yo, rtol=0, atol=TOL_Y)
    # detection is unaffected by the hook (nothing is written)
    assert eng.detectWatermark(dev(torch, yo), wm.MASK_TYPE.ME) == pytest.approx(O.detect(yo, W)[1], abs=TOL_CORR)
    eng.close()


@pytest.mark.parametrize("one_launch", [0, 1])
@pytest.mark.parametrize("shape", [(5, 260), (130, 516), (1080, 1920), (2160, 3840)])
@pytest.mark.parametrize("mask", ["ME", "NVF"])
@pytest.mark.parametrize("dtype", ["f32", "u8"])
def test_one_call_pair_equals_the_two_calls(wm, tc, shape, mask, dtype, one_launch, monkeypatch):
    """wm_embed_detect (makeWatermark + detectWatermark of its result as one call: both fused launches back to back, one wait)
    delivers what the two calls deliver -- bit for bit on the fused kernels; on the sweeps y and the strength bit for bit and
    the score to the rounding of the Gram sums' grouping (there the embed hands its output's lag sums to the detector,
    DESIGN 3c) -- and, through the two calls, what the oracle says; in place (the video contract) as well.
    one_launch: WM_FUSED_PAIR=1, both halves in ONE launch (k_fused_pair; off by default: it is not faster) -- the
    same bits again, also with a base that is not the input"""
    torch = tc
    monkeypatch.setenv("WM_FUSED_PAIR", str(one_launch))
    R, Cc = shape
    mk, omk = (wm.MASK_TYPE.ME, O.MASK_ME) if mask == "ME" else (wm.MASK_TYPE.NVF, O.MASK_NVF)
    x = synth_frame(R, Cc, frame=4)
    if dtype == "u8":
        x = np.floor(x).astype(np.uint8)
    W = synth_watermark(R, Cc)
    ef, es = engines(wm, R, Cc, W)
    xd = dev(torch, x)
    for eng in (ef, es):
        if eng is ef:
            eng.prof_enable(True)
        y2, a2 = eng.makeWatermark(xd, xd, mk)
        c2 = eng.detectWatermark(y2, mk)
        ctol = 0.0 if eng is ef else 2e-7
        for _ in range(3):
            y1, a1, c1 = eng.makeAndDetect(xd, xd, mk)
            assert a1 == a2 and abs(c1 - c2) <= ctol and torch.equal(y1, y2)
        frame = xd.clone()
        y1, a1, c1 = eng.makeAndDetect(frame, frame, mk, out=frame)
        assert a1 == a2 and abs(c1 - c2) <= ctol and torch.equal(frame, y2)
    rep = ef.prof_report()
    assert "k_fused_embed" in rep and "k_fused_detect" in rep and "k_gram" not in rep, rep
    assert ("k_fused_pair" in rep) == bool(one_launch), rep
    assert ef.fused_info()[3] == 0
    if R <= 200:
        # a grey base that is not the input (the pair kernel's other instance), and an unsolvable frame (constant: out = base,
        # the detector then scores the base)
        base = dev(torch, synth_frame(R, Cc, frame=9) if dtype == "f32" else np.floor(synth_frame(R, Cc, frame=9)).astype(np.uint8))
        yb, ab = ef.makeWatermark(xd, base, mk)
        cb = ef.detectWatermark(yb, mk)
        y1, a1b, c1b = ef.makeAndDetect(xd, base, mk)
        assert a1b == ab and c1b == cb and torch.equal(y1, yb)
        if mask == "ME":
            flat = torch.full_like(xd, 7)
            yf_, af_, cf_ = ef.makeAndDetect(flat, base, mk)
            assert af_ is None and torch.equal(yf_, base) and cf_ == ef.detectWatermark(base, mk)
        y1, a1, c1 = ef.makeAndDetect(xd, xd, mk)
    if dtype == "f32" and R * Cc <= 1080 * 1920:
        so, yo, ao = O.embed(x, x, W, mask=omk)
        assert a1 == pytest.approx(ao, rel=TOL_A)
        np.testing.assert_allclose(y1.cpu().numpy(), yo, rtol=0, atol=TOL_Y)
        assert c1 == pytest.approx(O.detect(yo, W, mask=omk)[1], abs=10 * TOL_CORR)
    ef.close(); es.close()


@pytest.mark.parametrize("dtype", ["f32", "u8"])
def test_one_launch_pair_is_stable_in_place(wm, tc, dtype, monkeypatch):
    """k_fused_pair's detector half reads y at addresses where -- in an in-place call -- the same launch read x: its halo loads
    go past the caches that are not coherent across the chip.  3 000 calls alternating in place / out of place: every strength
    and score bit-identical to the two separate calls (a 440 000-call run of the same loop over four shapes was clean)"""
    torch = tc
    monkeypatch.setenv("WM_FUSED_PAIR", "1")
    R, Cc = (1078, 1918) if dtype == "f32" else (1080, 1920)
    x = synth_frame(R, Cc, frame=2)
    if dtype == "u8":
        x = np.floor(x).astype(np.uint8)
    xd = dev(torch, x)
    eng = wm.Watermark(R, Cc, synth_watermark(R, Cc), 3, 40.0)
    y0, a0 = eng.makeWatermark(xd, xd, wm.MASK_TYPE.ME)
    c0 = eng.detectWatermark(y0, wm.MASK_TYPE.ME)
    frame = xd.clone()
    for i in range(3000):
        if i % 2:
            frame.copy_(xd)
            y, a, c = eng.makeAndDetect(frame, frame, wm.MASK_TYPE.ME, out=frame)
        else:
            y, a, c = eng.makeAndDetect(xd, xd, wm.MASK_TYPE.ME)
        assert a == a0 and c == c0, (i, a, a0, c, c0)
        if i % 500 < 2:
            assert torch.equal(y, y0)
    assert eng.fused_info()[3] == 0
    eng.close()


def test_one_call_pair_host_planes_batches_and_slots(wm, tc):
    """the one-call pair outside the fused case: host planes (staged in, the output staged out, the detector reads the
    device copy), a batch of frames, and a slot in flight (results at wm_sync); RGB output is refused"""
    torch = tc
    R, Cc = 270, 512
    W = synth_watermark(R, Cc)
    L = wm.lib()
    eng = wm.Watermark(R, Cc, W, 3, 40.0, nslots=2, max_frames=4)
    x = synth_frame(R, Cc, frame=3)
    xd = dev(torch, x)
    y2, a2 = eng.makeWatermark(xd, xd, wm.MASK_TYPE.ME)
    c2 = eng.detectWatermark(y2, wm.MASK_TYPE.ME)
    xh = np.ascontiguousarray(x)
    yh = np.empty_like(xh)

    def hp(a):
        return wm.wm_plane(a.ctypes.data, R, Cc, 1, wm.WM_F32, wm.WM_MEM_HOST, 1, Cc, 0, 0)
    av, cv, st = (C.c_float * 4)(), (C.c_float * 4)(), (C.c_int * 4)()
    pin, pout = hp(xh), hp(yh)
    assert L.wm_embed_detect(eng._ctx, 0, C.byref(pin), C.byref(pin), C.byref(pout), av, cv, st, wm.WM_SLOT_SYNC) == 0
    assert av[0] == a2 and cv[0] == c2 and st[0] == 0
    np.testing.assert_array_equal(yh, y2.cpu().numpy())
    # a batch (the sweeps) synchronously, then on a slot in flight
    xb = dev(torch, np.stack([synth_frame(R, Cc, frame=f) for f in range(4)]))
    yb2, ab2 = eng.makeWatermark(xb, xb, wm.MASK_TYPE.ME)
    cb2 = eng.detectWatermark(yb2, wm.MASK_TYPE.ME)
    yb1, ab1, cb1 = eng.makeAndDetect(xb, xb, wm.MASK_TYPE.ME)
    assert ab1 == ab2 and np.abs(np.array(cb1) - np.array(cb2)).max() <= 2e-7 and torch.equal(yb1, yb2)
    yb = torch.empty_like(xb)
    pb, po = wm.plane_of(xb, 1), wm.plane_of(yb, 1)
    torch.cuda.synchronize()
    assert L.wm_embed_detect(eng._ctx, 0, C.byref(pb), C.byref(pb), C.byref(po), av, cv, st, 1) == 0
    assert L.wm_sync(eng._ctx, 1) == 0
    assert list(av) == ab2 and np.abs(np.array(cv) - np.array(cb2)).max() <= 2e-7 and torch.equal(yb, yb2)
    assert "k_gram_ho" in eng.prof_report() if eng.prof_report() else True
    # RGB output: the detector would need the grey of the result -- refused, nothing queued
    rgb = torch.stack([xd, xd, xd])
    prgb = wm.plane_of(rgb, 3)
    orgb = wm.plane_of(torch.empty_like(rgb), 3)
    px = wm.plane_of(xd, 1)
    assert L.wm_embed_detect(eng._ctx, 0, C.byref(px), C.byref(prgb), C.byref(orgb), av, cv, st, wm.WM_SLOT_SYNC) == wm.WM_ERR_BAD_ARG
    eng.close()


@pytest.mark.parametrize("one_launch", [0, 1])
def test_one_call_pair_when_the_fused_embed_does_not_complete(wm, tc, monkeypatch, one_launch):
    """the time-out hooks under the one-call pair.  WM_FUSED_DBG=4 (a hand-off never completes, nothing is written): both
    operations are redone on the sweeps and answer correctly, also in place.  WM_FUSED_DBG=8 (output stores issued, end of
    the embed not observed): out of place the pair is redone on the sweeps; in place it must fail, never watermark twice."""
    torch = tc
    monkeypatch.setenv("WM_FUSED_PAIR", str(one_launch))   # (1: both halves in one launch, k_fused_pair: the same rules)
    R, Cc = 130, 516
    x = synth_frame(R, Cc, frame=2)
    W = synth_watermark(R, Cc)
    xd = dev(torch, x)
    so, yo, ao = O.embed(x, x, W, mask=O.MASK_ME)
    co = O.detect(yo, W, mask=O.MASK_ME)[1]
    monkeypatch.setenv("WM_FUSED_DBG", "4")
    eng = wm.Watermark(R, Cc, W, 3, 40.0)
    assert eng.fused_info()[0]
    frame = xd.clone()
    for k in range(12):   # call 0 times out, calls 1..8 sit in the back-off window (sweeps), call 9 probes again
        y, a, c = eng.makeAndDetect(frame, frame, wm.MASK_TYPE.ME, out=frame) if k == 0 else eng.makeAndDetect(xd, xd, wm.MASK_TYPE.ME)
        assert a == pytest.approx(ao, rel=TOL_A) and c == pytest.approx(co, abs=10 * TOL_CORR)
        np.testing.assert_allclose(y.cpu().numpy(), yo, rtol=0, atol=TOL_Y)
    assert eng.fused_info()[3] >= 2, eng.fused_info()
    eng.close()
    monkeypatch.setenv("WM_FUSED_DBG", "8")
    eng = wm.Watermark(R, Cc, W, 3, 40.0)
    y, a, c = eng.makeAndDetect(xd, xd, wm.MASK_TYPE.ME)
    assert a == pytest.approx(ao, rel=TOL_A) and c == pytest.approx(co, abs=10 * TOL_CORR)
    np.testing.assert_allclose(y.cpu().numpy(), yo, rtol=0, atol=TOL_Y)
    assert eng.fused_info()[3] == 1
    eng.close()
    eng = wm.Watermark(R, Cc, W, 3, 40.0)
    frame = xd.clone()
    with pytest.raises(RuntimeError, match="completion of the output stores was not observed"):
        eng.makeAndDetect(frame, frame, wm.MASK_TYPE.ME, out=frame)
    np.testing.assert_allclose(frame.cpu().numpy(), yo, rtol=0, atol=TOL_Y)
    # the context is usable afterwards
    assert eng.detectWatermark(dev(torch, yo), wm.MASK_TYPE.ME) == pytest.approx(co, abs=TOL_CORR)
    eng.close()


CHILD_XPROC = r"""
import importlib, json, os, sys
import numpy as np
sys.path.insert(0, os.environ["WM_ROOT"]); sys.path.insert(0, os.path.join(os.environ["WM_ROOT"], "tests"))
import torch
wm = importlib.import_module("watermarking-gpu_amd")
from synth import synth_frame, synth_watermark
R, C = 1080, 1920
k = int(os.environ["WM_CHILD_K"])
W = synth_watermark(R, C)
x = torch.from_numpy(synth_frame(R, C, frame=k)).cuda()
eng = wm.Watermark(R, C, W, 3, 40.0)
assert eng.fused_info()[0]
y0, a0 = eng.makeWatermark(x, x, wm.MASK_TYPE.ME)
c0 = eng.detectWatermark(y0, wm.MASK_TYPE.ME)
print("READY", flush=True)
sys.stdin.readline()                                   # both processes start their loops together
bad = 0
for _ in range(int(os.environ["WM_CHILD_LOOPS"])):
    y, a = eng.makeWatermark(x, x, wm.MASK_TYPE.ME)
    c = eng.detectWatermark(y, wm.MASK_TYPE.ME)
    bad += int(a != a0 or c != c0)
bad += int(not torch.equal(y, y0))
print("RESULT " + json.dumps({"fallbacks": eng.fused_info()[3], "bad": bad, "a": a0, "corr": c0}), flush=True)
eng.close()
"""


def test_fused_calls_from_two_processes_on_one_device(wm, tc):
    """two PROCESSES on one device, each issuing fused single-image calls as fast as it can: the per-device lock file
    (flock) serialises their fused launches like the mutex does for threads, so neither grid waits for workgroups the other
    grid keeps off the CUs -- no time-outs, no fallbacks, identical results"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    procs = []
    for k in range(2):
        env = dict(os.environ, WM_ROOT=root, WM_CHILD_K=str(k), WM_CHILD_LOOPS="300")
        procs.append(subprocess.Popen([sys.executable, "-c", CHILD_XPROC], env=env, stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    try:
        for p in procs:
            line = p.stdout.readline()
            assert line.startswith("READY"), line + p.stderr.read()[-3000:]
        for p in procs:
            p.stdin.write("go\n"); p.stdin.flush()
        for p in procs:
            out, err = p.communicate(timeout=300)
            assert p.returncode == 0, err[-3000:]
            rec = json.loads([l for l in out.splitlines() if l.startswith("RESULT ")][-1][7:])
            assert rec["bad"] == 0 and rec["fallbacks"] == 0, rec
            assert 0.2 < rec["corr"] < 1.0
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()


def test_fused_lock_held_by_a_stopped_process_does_not_hang_the_call(wm, tc, tmp_path, monkeypatch):
    """the per-device lock file is tried without blocking for a few milliseconds: while somebody else holds it (here: this
    test, on a descriptor of its own -- flock conflicts between open file descriptions, also inside one process) a
    synchronous one-image call runs on the sweeps, with the same results, and says so in wm_fused_lock_skips; once the lock
    is free the fused kernels take the calls again.  The file is opened read-only and never through a symbolic link."""
    import fcntl
    import glob
    import os
    import time
    monkeypatch.setenv("WM_FUSED_LOCK_DIR", str(tmp_path))
    R, Cc = 130, 516
    x = synth_frame(R, Cc, frame=1)
    W = synth_watermark(R, Cc)
    eng = wm.Watermark(R, Cc, W, 3, 40.0)
    assert eng.fused_info()[0]
    files = glob.glob(os.path.join(str(tmp_path), "wm_fused_*.lock"))
    assert len(files) == 1 and not os.path.islink(files[0])
    xd = dev(tc, x)
    y0, a0 = eng.makeWatermark(xd, xd, wm.MASK_TYPE.ME)
    c0 = eng.detectWatermark(y0, wm.MASK_TYPE.ME)
    L = wm.lib()
    assert L.wm_fused_lock_skips(eng._ctx) == 0
    fd = os.open(files[0], os.O_RDONLY)
    try:
        fcntl.flock(fd, fcntl.LOCK_EX)
        t0 = time.perf_counter()
        y1, a1 = eng.makeWatermark(xd, xd, wm.MASK_TYPE.ME)
        c1 = eng.detectWatermark(y1, wm.MASK_TYPE.ME)
        dt = time.perf_counter() - t0
        assert L.wm_fused_lock_skips(eng._ctx) == 2 and dt < 1.0
        assert abs(a1 - a0) <= 1e-6 * abs(a0) and abs(c1 - c0) <= 1e-6 and float((y1 - y0).abs().max()) <= 1e-4
        y2, a2, c2 = eng.makeAndDetect(xd, xd, wm.MASK_TYPE.ME)   # the pair as one call: on the sweeps as well, one more skip
        assert L.wm_fused_lock_skips(eng._ctx) == 4   # (its embed and its detector each tried the lock once)
        assert abs(a2 - a0) <= 1e-6 * abs(a0) and abs(c2 - c0) <= 2e-6
    finally:
        fcntl.flock(fd, fcntl.LOCK_UN)
        os.close(fd)
    skips = L.wm_fused_lock_skips(eng._ctx)
    y3, a3 = eng.makeWatermark(xd, xd, wm.MASK_TYPE.ME)
    assert L.wm_fused_lock_skips(eng._ctx) == skips and a3 == a0 and bool((y3 == y0).all())
    assert eng.fused_info()[3] == 0  # no time-out fallbacks anywhere
    eng.close()
