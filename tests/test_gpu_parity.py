"""GPU parity tests: the HIP path, called through the C ABI (ctypes mirror), against the CPU oracle on the
same inputs.  Tolerances (GPU vs oracle): coefficients <= 1e-4 abs, a <= 1e-4 rel,
correlation <= 1e-5 abs, NVF mask <= 1e-5 abs, e / ME mask given identical coefficients <= 1e-4 rel,
y <= 1e-3 abs (u8 output: <= 1 LSB on <= 0.1 % of the pixels); indexing, passthrough and run-to-run
determinism bit-exact."""
import numpy as np
import pytest

import oracle_lib as O
from synth import synth_frame, synth_watermark

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["fused", "sweeps"])
def single_call_path(request, monkeypatch):
    """every test runs twice: synchronous one-frame calls on the fused single-launch kernels (the default) and on the
    batched sweeps (WM_FUSED=0, read when an engine is created)"""
    monkeypatch.setenv("WM_FUSED", "1" if request.param == "fused" else "0")
    return request.param

TOL_C, TOL_A, TOL_CORR, TOL_NVF, TOL_Y = 1e-4, 1e-4, 1e-5, 1e-5, 1e-3
TOL_C_EXACT = 2e-7  # what the exact-f64 Gram kernel actually achieves (one f32 ulp of the coefficients)

SHAPES = [(64, 64), (70, 131), (96, 200), (128, 256), (130, 260), (257, 515), (300, 1030)]


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


def dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def make(wm, shape, p=3, psnr=40.0, frame=0):
    x = synth_frame(shape[0], shape[1], frame=frame)
    W = synth_watermark(shape[0], shape[1])
    eng = wm.Watermark(shape[0], shape[1], W, p, psnr)
    return x, W, eng


def test_native_library_is_loaded(wm, torch_cuda):
    L = wm.lib()
    assert b"gfx950" in L.wm_version()
    import ctypes
    assert isinstance(L, ctypes.CDLL)


@pytest.mark.parametrize("shape", [(1, 1), (2, 3), (3, 5), (4, 5), (4, 6), (5, 5), (9, 4), (20, 5), (7, 9), (3, 700)] + SHAPES)
@pytest.mark.parametrize("dtype", ["f32", "u8"])
def test_gram_exact(wm, torch_cuda, shape, dtype):
    """k_gram (13 f64 lag sums over the core + exact border frame) against the oracle's f64 Gram:
    both sum exact products in f64, so they agree to reduction-order rounding (1e-13 relative)"""
    torch = torch_cuda
    x = synth_frame(shape[0], shape[1], frame=2, dtype=np.uint8 if dtype == "u8" else np.float32)
    eng = wm.Watermark(shape[0], shape[1], synth_watermark(*shape), 3, 40.0)
    Rx, rx = eng.gram(dev(torch, x))
    Ro, ro = O.gram(x.astype(np.float32))
    np.testing.assert_allclose(Rx, Ro, rtol=1e-13, atol=0)
    np.testing.assert_allclose(rx, ro, rtol=1e-13, atol=0)


def test_gram_model_matches_oracle():
    """the numpy model of the lag/border split (tests/lag_gram_model.py) against the oracle"""
    import lag_gram_model as LG
    x = synth_frame(9, 13, frame=1)
    Rx, rx = LG.gram_by_lags(x)
    Ro, ro = O.gram(x)
    np.testing.assert_allclose(Rx, Ro, rtol=1e-13)
    np.testing.assert_allclose(rx, ro, rtol=1e-13)


@pytest.mark.parametrize("shape", SHAPES)
def test_coefficients_and_me_mask(wm, torch_cuda, shape):
    torch = torch_cuda
    x, W, eng = make(wm, shape)
    m, e, c, st = eng.computeMask(dev(torch, x), wm.MASK_TYPE.ME, want_error_sequence=True)
    so, co, eo, mo, mxo = O.me_mask(x)
    assert st == 0 and so == 0
    np.testing.assert_allclose(c, co, rtol=0, atol=TOL_C_EXACT)
    # error sequence / mask: compare given IDENTICAL coefficients (the GPU's)
    e_ref = O.error_sequence(x, c)
    np.testing.assert_array_equal(e.cpu().numpy(), e_ref)  # same f32 op order => bit-exact
    ae = np.abs(e_ref)
    np.testing.assert_array_equal(m.cpu().numpy(), ae / ae.max())
    np.testing.assert_allclose(m.cpu().numpy(), mo, rtol=0, atol=1e-4)


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("p", [3, 5, 7, 9])
def test_nvf_mask(wm, torch_cuda, shape, p):
    torch = torch_cuda
    x, W, eng = make(wm, shape, p=p)
    m, _, _, st = eng.computeMask(dev(torch, x), wm.MASK_TYPE.NVF)
    ref = O.nvf_mask(x, p)
    got = m.cpu().numpy()
    np.testing.assert_allclose(got, ref, rtol=0, atol=TOL_NVF)
    np.testing.assert_array_equal(got, ref)  # pinned op order: bit-exact


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("mask", ["ME", "NVF"])
def test_embed_detect_f32(wm, torch_cuda, shape, mask):
    torch = torch_cuda
    x, W, eng = make(wm, shape)
    mt = wm.MASK_TYPE[mask]
    y, a = eng.makeWatermark(dev(torch, x), dev(torch, x), mt)
    so, yo, ao = O.embed(x, x, W, mask=int(mt))
    assert a == pytest.approx(ao, rel=TOL_A)
    yg = y.cpu().numpy()
    np.testing.assert_allclose(yg, yo, rtol=0, atol=TOL_Y)
    # detector on the GPU's own output and on the oracle's output
    corr = eng.detectWatermark(y, mt)
    s2, corr_o = O.detect(yg, W, mask=int(mt))
    assert corr == pytest.approx(corr_o, abs=TOL_CORR)
    corr2 = eng.detectWatermark(dev(torch, yo), mt)
    s3, corr_o2 = O.detect(yo, W, mask=int(mt))
    assert corr2 == pytest.approx(corr_o2, abs=TOL_CORR)
    # unmarked image: correlation near zero, and equal to the oracle's
    c0 = eng.detectWatermark(dev(torch, x), mt)
    assert c0 == pytest.approx(O.detect(x, W, mask=int(mt))[1], abs=TOL_CORR)
    assert corr > 0.2 > abs(c0)


@pytest.mark.parametrize("p", [5, 7, 9])
def test_nvf_embed_detect_larger_windows(wm, torch_cuda, p):
    torch = torch_cuda
    shape = (130, 520)
    x, W, eng = make(wm, shape, p=p)
    y, a = eng.makeWatermark(dev(torch, x), dev(torch, x), wm.MASK_TYPE.NVF)
    so, yo, ao = O.embed(x, x, W, p=p, mask=O.MASK_NVF)
    assert a == pytest.approx(ao, rel=TOL_A)
    np.testing.assert_allclose(y.cpu().numpy(), yo, rtol=0, atol=TOL_Y)
    corr = eng.detectWatermark(dev(torch, yo), wm.MASK_TYPE.NVF)
    assert corr == pytest.approx(O.detect(yo, W, p=p, mask=O.MASK_NVF)[1], abs=TOL_CORR)
    with pytest.raises(RuntimeError):  # ME only for p = 3 (main.cpp:89)
        eng.makeWatermark(dev(torch, x), dev(torch, x), wm.MASK_TYPE.ME)


@pytest.mark.parametrize("tag", ["512", "720p_crop"])
def test_golden_fixtures(wm, torch_cuda, golden, tag):
    """harness flow of main.cpp:169-226 on the reference's sample data: embed on the RGB base, detect on grey"""
    from conftest import load_pair
    torch = torch_cuda
    rgb, W = load_pair(golden, tag)
    gray = O.rgb2gray(rgb)
    g = golden[tag]
    eng = wm.Watermark(gray.shape[0], gray.shape[1], W, 3, 40.0)
    m, e, c, st = eng.computeMask(dev(torch, gray), wm.MASK_TYPE.ME)
    np.testing.assert_allclose(c, g["coefficients"], rtol=0, atol=TOL_C_EXACT)
    for name in ("NVF", "ME"):
        mt = wm.MASK_TYPE[name]
        y, a = eng.makeWatermark(dev(torch, gray), dev(torch, gray), mt)
        assert a == pytest.approx(g[name]["a"], rel=TOL_A)
        assert eng.detectWatermark(y, mt) == pytest.approx(g[name]["corr_gray"], abs=TOL_CORR)
        yrgb, a2 = eng.makeWatermark(dev(torch, gray), dev(torch, rgb), mt)
        assert a2 == pytest.approx(g[name]["a"], rel=TOL_A)
        so, yrgb_o, _ = O.embed(gray, rgb, W, mask=int(mt))
        np.testing.assert_allclose(yrgb.cpu().numpy(), yrgb_o, rtol=0, atol=TOL_Y)
        gw = O.rgb2gray(yrgb.cpu().numpy())
        assert eng.detectWatermark(dev(torch, gw), mt) == pytest.approx(g[name]["corr_rgb_harness"], abs=TOL_CORR)
        assert eng.detectWatermark(dev(torch, gray), mt) == pytest.approx(g[name]["corr_unmarked"], abs=TOL_CORR)


def test_golden_crop_per_pixel(wm, torch_cuda, golden, pair_crop):
    import os
    from conftest import GOLDEN
    torch = torch_cuda
    rgb, W = pair_crop
    exp = np.load(os.path.join(GOLDEN, "720p_crop_expected.npz"))
    gray = exp["gray"]
    eng = wm.Watermark(gray.shape[0], gray.shape[1], W, 3, 40.0)
    m, _, _, _ = eng.computeMask(dev(torch, gray), wm.MASK_TYPE.NVF)
    np.testing.assert_array_equal(m.cpu().numpy(), exp["m_nvf"])
    m, e, c, _ = eng.computeMask(dev(torch, gray), wm.MASK_TYPE.ME, want_error_sequence=True)
    np.testing.assert_allclose(e.cpu().numpy(), exp["e"], rtol=0, atol=2e-3)
    np.testing.assert_allclose(m.cpu().numpy(), exp["m_me"], rtol=0, atol=1e-4)
    y, a = eng.makeWatermark(dev(torch, gray), dev(torch, gray), wm.MASK_TYPE.ME)
    np.testing.assert_allclose(y.cpu().numpy(), exp["y_me"], rtol=0, atol=TOL_Y)
    y, a = eng.makeWatermark(dev(torch, gray), dev(torch, gray), wm.MASK_TYPE.NVF)
    np.testing.assert_allclose(y.cpu().numpy(), exp["y_nvf"], rtol=0, atol=TOL_Y)


@pytest.mark.parametrize("shape", [(64, 64), (98, 300), (135, 514)])
@pytest.mark.parametrize("mask", ["ME", "NVF"])
def test_video_u8_frames(wm, torch_cuda, shape, mask):
    """Y plane u8 -> f32 -> makeWatermark(frame, frame) -> u8 by truncation (main.cpp:355-357,379-381,405)"""
    torch = torch_cuda
    x8 = synth_frame(shape[0], shape[1], frame=7, dtype=np.uint8)
    W = synth_watermark(shape[0], shape[1])
    eng = wm.Watermark(shape[0], shape[1], W, 3, 40.0)
    mt = wm.MASK_TYPE[mask]
    y, a = eng.makeWatermark(dev(torch, x8), dev(torch, x8), mt)
    so, yo, ao = O.embed_u8(x8, W, mask=int(mt))
    assert y.dtype == torch.uint8
    assert a == pytest.approx(ao, rel=TOL_A)
    diff = np.abs(y.cpu().numpy().astype(int) - yo.astype(int))
    assert diff.max() <= 1 and (diff != 0).mean() <= 1e-3
    corr = eng.detectWatermark(dev(torch, yo), mt)
    assert corr == pytest.approx(O.detect_u8(yo, W, mask=int(mt))[1], abs=TOL_CORR)
    # in-place (out aliases base and input), as the video path does
    xin = dev(torch, x8)
    y2, a2 = eng.makeWatermark(xin, xin, mt, out=xin)
    assert a2 == a
    np.testing.assert_array_equal(xin.cpu().numpy(), y.cpu().numpy())


def test_unsolvable_passthrough(wm, torch_cuda):
    """constant image => singular system => out = base bit-exact, strength unset, detect 0.0 (Watermark.cpp:164-165,246-247)"""
    torch = torch_cuda
    shape = (96, 300)
    x = np.full(shape, 117.0, np.float32)
    W = synth_watermark(*shape)
    base = synth_frame(*shape, frame=1)
    eng = wm.Watermark(shape[0], shape[1], W, 3, 40.0)
    y, a = eng.makeWatermark(dev(torch, x), dev(torch, base), wm.MASK_TYPE.ME)
    assert a is None
    np.testing.assert_array_equal(y.cpu().numpy(), base)
    rgb = np.stack([base, base + 1, base + 2]).astype(np.float32)
    y, a = eng.makeWatermark(dev(torch, x), dev(torch, rgb), wm.MASK_TYPE.ME)
    assert a is None
    np.testing.assert_array_equal(y.cpu().numpy(), rgb)
    assert eng.detectWatermark(dev(torch, x), wm.MASK_TYPE.ME) == 0.0
    assert eng.detectWatermark(dev(torch, x), wm.MASK_TYPE.NVF) == 0.0
    # the engine stays usable afterwards
    x2 = synth_frame(*shape)
    y, a = eng.makeWatermark(dev(torch, x2), dev(torch, x2), wm.MASK_TYPE.ME)
    assert a == pytest.approx(O.embed(x2, x2, W)[2], rel=TOL_A)


def test_batched_frames_equal_single_frames(wm, torch_cuda):
    torch = torch_cuda
    shape, F = (120, 300), 5
    W = synth_watermark(*shape)
    frames = np.stack([synth_frame(*shape, frame=f) for f in range(F)])
    frames[3] = 42.0  # one unsolvable frame in the middle of the batch
    eng = wm.Watermark(shape[0], shape[1], W, 3, 40.0, nslots=2, max_frames=F)
    xb = dev(torch, frames)
    yb, ab = eng.makeWatermark(xb, xb, wm.MASK_TYPE.ME)
    cb = eng.detectWatermark(yb, wm.MASK_TYPE.ME)
    for f in range(F):
        y1, a1 = eng.makeWatermark(xb[f], xb[f], wm.MASK_TYPE.ME)
        assert a1 == ab[f]
        np.testing.assert_array_equal(y1.cpu().numpy(), yb[f].cpu().numpy())
        assert abs(eng.detectWatermark(y1, wm.MASK_TYPE.ME) - cb[f]) <= 1.2e-7  # (partial sums grouped by another tiling: last bit)
    assert ab[3] is None and cb[3] == 0.0
    np.testing.assert_array_equal(yb[3].cpu().numpy(), frames[3])


def test_pitched_and_unaligned_planes(wm, torch_cuda):
    """row pitch > cols (FFmpeg linesize, main.cpp:348-353) and a base pointer that is not 16-byte aligned"""
    torch = torch_cuda
    shape = (90, 300)
    x, W, eng = make(wm, shape)
    so, yo, ao = O.embed(x, x, W)
    big = torch.zeros((shape[0], 333), dtype=torch.float32, device="cuda")
    big[:, 1:301] = dev(torch, x)
    view = big[:, 1:301]  # pitch 333, base offset 4 bytes
    y, a = eng.makeWatermark(view, view, wm.MASK_TYPE.ME)
    assert a == pytest.approx(ao, rel=TOL_A)
    np.testing.assert_allclose(y.cpu().numpy(), yo, rtol=0, atol=TOL_Y)
    out = torch.zeros_like(big)
    y2, a2 = eng.makeWatermark(view, view, wm.MASK_TYPE.ME, out=out[:, 2:302])
    np.testing.assert_array_equal(out[:, 2:302].cpu().numpy(), y.cpu().numpy())
    assert float(out[:, :2].abs().sum()) == 0.0 and float(out[:, 302:].abs().sum()) == 0.0  # no stray writes
    assert eng.detectWatermark(out[:, 2:302], wm.MASK_TYPE.ME) == pytest.approx(O.detect(yo, W)[1], abs=TOL_CORR)


def test_determinism_and_slots(wm, torch_cuda):
    torch = torch_cuda
    shape = (200, 520)
    x, W, eng = make(wm, shape)
    xd = dev(torch, x)
    y1, a1 = eng.makeWatermark(xd, xd, wm.MASK_TYPE.ME)
    y2, a2 = eng.makeWatermark(xd, xd, wm.MASK_TYPE.ME)
    assert a1 == a2
    assert torch.equal(y1, y2)
    c1 = eng.detectWatermark(y1, wm.MASK_TYPE.ME)
    assert c1 == eng.detectWatermark(y2, wm.MASK_TYPE.ME)
    # two frames in flight on two slots give the same answers as the synchronous calls
    import ctypes as C
    outs = [torch.empty_like(xd), torch.empty_like(xd)]
    a = [(C.c_float * 1)(), (C.c_float * 1)()]
    corr = [(C.c_float * 1)(), (C.c_float * 1)()]
    torch.cuda.synchronize()
    for s in range(2):
        eng.embed_async(xd, xd, outs[s], wm.MASK_TYPE.ME, s, a_out=a[s])
        eng.detect_async(outs[s], wm.MASK_TYPE.ME, s, corr_out=corr[s])
    for s in range(2):
        assert eng.sync(s) == 0
        assert a[s][0] == a1 and corr[s][0] == c1
        assert torch.equal(outs[s], y1)


def test_copy_and_reinitialize(wm, torch_cuda):
    torch = torch_cuda
    x, W, eng = make(wm, (80, 260))
    xd = dev(torch, x)
    y, a = eng.makeWatermark(xd, xd, wm.MASK_TYPE.NVF)
    eng2 = eng.copy()  # shares W (Watermark.cpp:30-37)
    y2, a2 = eng2.makeWatermark(xd, xd, wm.MASK_TYPE.NVF)
    assert a2 == a and torch.equal(y, y2)
    x3 = synth_frame(70, 131)
    W3 = synth_watermark(70, 131)
    eng.reinitialize(W3, 70, 131)  # Watermark.cpp:78-85
    y3, a3 = eng.makeWatermark(dev(torch, x3), dev(torch, x3), wm.MASK_TYPE.ME)
    assert a3 == pytest.approx(O.embed(x3, x3, W3)[2], rel=TOL_A)
    with pytest.raises(RuntimeError):  # plane of the old size is rejected
        eng.makeWatermark(xd, xd, wm.MASK_TYPE.ME)
    y2b, a2b = eng2.makeWatermark(xd, xd, wm.MASK_TYPE.NVF)  # the copy is unaffected
    assert a2b == a


def test_w_file_loading(wm, torch_cuda, tmp_path):
    """W(r,c) = file[r*cols + c] -- bit-exact indexing (Watermark.cpp:62-75)"""
    torch = torch_cuda
    shape = (64, 96)
    W = synth_watermark(*shape)
    f = tmp_path / "w.dat"
    W.tofile(f)
    eng = wm.Watermark(shape[0], shape[1], str(f), 3, 40.0)
    x = synth_frame(*shape)
    y, a = eng.makeWatermark(dev(torch, x), dev(torch, x), wm.MASK_TYPE.NVF)
    assert a == pytest.approx(O.embed(x, x, W, mask=O.MASK_NVF)[2], rel=1e-6)
    with pytest.raises(RuntimeError, match="W file total elements"):
        wm.Watermark(shape[0] + 1, shape[1], str(f), 3, 40.0)
    with pytest.raises(RuntimeError, match="Error opening"):
        wm.Watermark(shape[0], shape[1], str(tmp_path / "missing.dat"), 3, 40.0)
    with pytest.raises(RuntimeError, match="p parameter"):
        wm.Watermark(shape[0], shape[1], W, 4, 40.0)


@pytest.mark.parametrize("rps", [1, 3, 8, 17, 64])
def test_rows_per_segment_invariance(wm, torch_cuda, rps):
    """segment length is a tuning knob: results may move only by reduction-order noise"""
    torch = torch_cuda
    shape = (75, 300)
    x, W, eng = make(wm, shape)
    eng.set_rows_per_segment(rps)
    xd = dev(torch, x)
    y, a = eng.makeWatermark(xd, xd, wm.MASK_TYPE.ME)
    so, yo, ao = O.embed(x, x, W)
    assert a == pytest.approx(ao, rel=TOL_A)
    np.testing.assert_allclose(y.cpu().numpy(), yo, rtol=0, atol=TOL_Y)
    assert eng.detectWatermark(dev(torch, yo), wm.MASK_TYPE.ME) == pytest.approx(O.detect(yo, W)[1], abs=TOL_CORR)
    m, _, _, _ = eng.computeMask(xd, wm.MASK_TYPE.NVF)
    np.testing.assert_array_equal(m.cpu().numpy(), O.nvf_mask(x))


@pytest.mark.parametrize("cols", [257, 258, 259, 513, 514])
@pytest.mark.parametrize("p", [3, 9])
def test_few_columns_right_of_a_full_strip(wm, torch_cuda, cols, p):
    """aligned (pitched) planes whose width leaves 1..3 columns right of the last full 256-column strip: the aligned
    path's vector halo load would straddle the image edge, so that strip must take the generic path"""
    torch = torch_cuda
    R, pitch = 40, ((cols + 3) // 4) * 4 + 4
    x = synth_frame(R, cols, frame=3)
    W = synth_watermark(R, cols)
    eng = wm.Watermark(R, cols, W, p, 40.0)
    big = torch.full((R + 1, pitch), 1e6, dtype=torch.float32, device="cuda")   # poison in the pitch padding
    big[:R, :cols] = dev(torch, x)
    view = big[:R, :cols]
    m, _, _, _ = eng.computeMask(view, wm.MASK_TYPE.NVF)
    np.testing.assert_array_equal(m.cpu().numpy(), O.nvf_mask(x, p))
    y, a = eng.makeWatermark(view, view, wm.MASK_TYPE.NVF)
    so, yo, ao = O.embed(x, x, W, p=p, mask=O.MASK_NVF)
    assert a == pytest.approx(ao, rel=TOL_A)
    np.testing.assert_allclose(y.cpu().numpy(), yo, rtol=0, atol=TOL_Y)
    assert eng.detectWatermark(y, wm.MASK_TYPE.NVF) == pytest.approx(O.detect(yo, W, p=p, mask=O.MASK_NVF)[1], abs=TOL_CORR)
    if p == 3:
        Rx, rx = eng.gram(view)
        Ro, ro = O.gram(x)
        np.testing.assert_allclose(Rx, Ro, rtol=1e-13)
        y, a = eng.makeWatermark(view, view, wm.MASK_TYPE.ME)
        so, yo, ao = O.embed(x, x, W)
        assert a == pytest.approx(ao, rel=TOL_A)
        np.testing.assert_allclose(y.cpu().numpy(), yo, rtol=0, atol=TOL_Y)
        assert eng.detectWatermark(view, wm.MASK_TYPE.ME) == pytest.approx(O.detect(x, W)[1], abs=TOL_CORR)


@pytest.mark.parametrize("cols", [260, 300, 508, 764])
@pytest.mark.parametrize("dtype", ["f32", "u8"])
def test_shifted_last_strip(wm, torch_cuda, cols, dtype, single_call_path):
    """widths that are a multiple of 4 but not of 256: the aligned path moves the last, partial strip left so that it
    ends at the last column (1920 = 7 strips + one shifted by 128); its leading lanes duplicate pixels of the previous
    strip and must neither be summed twice nor stored.  Checked: exact Gram, masks, embed (also in place and with an RGB
    base), detect, against the oracle; and that the result equals the generic path's (an unaligned view) bit for bit."""
    torch = torch_cuda
    R = 70
    npdt = np.float32 if dtype == "f32" else np.uint8
    x = synth_frame(R, cols, frame=6, dtype=npdt)
    W = synth_watermark(R, cols)
    eng = wm.Watermark(R, cols, W, 3, 40.0)
    xd = dev(torch, x)
    Rx, rx = eng.gram(xd)
    Ro, ro = O.gram(x.astype(np.float32))
    np.testing.assert_allclose(Rx, Ro, rtol=1e-13)
    np.testing.assert_allclose(rx, ro, rtol=1e-13)
    xf = x.astype(np.float32)
    for mk, omk in ((wm.MASK_TYPE.ME, O.MASK_ME), (wm.MASK_TYPE.NVF, O.MASK_NVF)):
        y, a = eng.makeWatermark(xd, xd, mk)
        if dtype == "f32":
            so, yo, ao = O.embed(xf, xf, W, mask=omk)
            np.testing.assert_allclose(y.cpu().numpy(), yo, rtol=0, atol=TOL_Y)
            cref = O.detect(yo, W, mask=omk)[1]
        else:
            so, yo, ao = O.embed_u8(x, W, mask=omk)
            d = np.abs(y.cpu().numpy().astype(int) - yo.astype(int))
            assert d.max() <= 1 and (d != 0).mean() <= 1e-3
            cref = O.detect_u8(yo, W, mask=omk)[1]
        assert a == pytest.approx(ao, rel=TOL_A)
        assert eng.detectWatermark(dev(torch, yo), mk) == pytest.approx(cref, abs=TOL_CORR)
        # the same frame through the generic path: a view whose base is off the vector alignment
        big = torch.zeros((R, cols + 8), dtype=xd.dtype, device="cuda")
        big[:, 1:cols + 1] = xd
        view = big[:, 1:cols + 1]
        y2, a2 = eng.makeWatermark(view, view, mk)
        if single_call_path == "sweeps":
            np.testing.assert_array_equal(y2.cpu().numpy(), y.cpu().numpy())
            assert a2 == a
        else:
            assert a2 == pytest.approx(a, rel=1e-6)
            assert np.abs(y2.cpu().numpy().astype(np.float64) - y.cpu().numpy()).max() <= (1 if dtype == "u8" else 2e-4)
        # (the aligned frame takes the fused kernels when they are on: another grouping of the same partial sums)
        assert eng.detectWatermark(view, mk) == pytest.approx(eng.detectWatermark(xd, mk), abs=0 if single_call_path == "sweeps" else 2e-6)
        # in place, as the video path does
        xin = xd.clone()
        y3, a3 = eng.makeWatermark(xin, xin, mk, out=xin)
        np.testing.assert_array_equal(xin.cpu().numpy(), y.cpu().numpy())
    if dtype == "f32":
        rgb = np.stack([synth_frame(R, cols, frame=10 + k) for k in range(3)])
        yrgb, a = eng.makeWatermark(xd, dev(torch, rgb), wm.MASK_TYPE.ME)
        so, yo, ao = O.embed(xf, rgb, W)
        np.testing.assert_allclose(yrgb.cpu().numpy(), yo, rtol=0, atol=TOL_Y)
    # materialised masks (k_mask stores per lane: the duplicate lanes must not write)
    m, e, c, st = eng.computeMask(dev(torch, xf), wm.MASK_TYPE.ME, want_error_sequence=True)
    e_ref = O.error_sequence(xf, c)
    np.testing.assert_array_equal(e.cpu().numpy(), e_ref)
    np.testing.assert_array_equal(m.cpu().numpy(), np.abs(e_ref) / np.abs(e_ref).max())
    mn, _, _, _ = eng.computeMask(dev(torch, xf), wm.MASK_TYPE.NVF)
    np.testing.assert_array_equal(mn.cpu().numpy(), O.nvf_mask(xf))


def test_random_shapes_against_oracle(wm, torch_cuda):
    """seeded sweep over shapes, element types, masks, segment lengths and batch sizes: every strip / segment / edge / quad combination the
    launch geometry can produce (full strips, shifted last strip, ragged generic strip, 1..3 columns left over, partial
    last segment, one-block images) against the oracle"""
    torch = torch_cuda
    rng = np.random.default_rng(20240607)
    for case in range(28):
        R = int(rng.integers(64, 400))
        Cc = int(rng.choice([rng.integers(64, 900), 4 * rng.integers(16, 225), 256 * rng.integers(1, 4) + rng.integers(0, 8)]))
        u8 = bool(rng.integers(0, 2))
        mk, omk = (wm.MASK_TYPE.ME, O.MASK_ME) if rng.integers(0, 3) else (wm.MASK_TYPE.NVF, O.MASK_NVF)
        F = int(rng.integers(1, 8))  # frames per launch: 4 and more take the frame-quad block mapping
        xs = np.stack([synth_frame(R, Cc, frame=8 * case + f, dtype=np.uint8 if u8 else np.float32) for f in range(F)])
        W = synth_watermark(R, Cc)
        eng = wm.Watermark(R, Cc, W, 3, 40.0, nslots=1, max_frames=F)
        if rng.integers(0, 2):
            eng.set_rows_per_segment(int(rng.integers(5, 70)))
        xd = dev(torch, xs)
        ys, as_ = eng.makeWatermark(xd, xd, mk)
        k = int(rng.integers(0, F))  # one frame of the batch against the oracle
        x, y, a = xs[k], ys[k], as_[k]
        tag = f"case {case}: {R}x{Cc} {'u8' if u8 else 'f32'} mask={int(mk)} F={F} frame {k}"
        if u8:
            so, yo, ao = O.embed_u8(x, W, mask=omk)
            d = np.abs(y.cpu().numpy().astype(int) - yo.astype(int))
            assert d.max() <= 1 and (d != 0).mean() <= 2e-3, tag
            cref = O.detect_u8(yo, W, mask=omk)[1]
        else:
            so, yo, ao = O.embed(x, x, W, mask=omk)
            np.testing.assert_allclose(y.cpu().numpy(), yo, rtol=0, atol=TOL_Y, err_msg=tag)
            cref = O.detect(yo, W, mask=omk)[1]
        assert a == pytest.approx(ao, rel=TOL_A), tag
        yb = ys.clone()
        yb[k] = dev(torch, yo)
        assert eng.detectWatermark(yb, mk)[k] == pytest.approx(cref, abs=TOL_CORR), tag
        eng.close()


@pytest.mark.parametrize("rows", [33, 40, 100, 196])
@pytest.mark.parametrize("mask", ["ME", "NVF"])
def test_short_wide_batches_fill_the_record_arrays(wm, torch_cuda, rows, mask):
    """frames == max_frames >= 16 on short, wide images: make_geom's balancing step shortens the segments below 8 rows
    (rows = 100: 7-row segments, 15 per strip against ceil(100 / 8) = 13), so the per-wave record arrays must be sized by
    the balanced segment count; every frame of the full batch against the oracle (a and corr, first and last frame: y)"""
    torch = torch_cuda
    Cc, F = 2560, 16
    mk, omk = (wm.MASK_TYPE.ME, O.MASK_ME) if mask == "ME" else (wm.MASK_TYPE.NVF, O.MASK_NVF)
    xs = np.stack([synth_frame(rows, Cc, frame=f) for f in range(F)])
    W = synth_watermark(rows, Cc)
    eng = wm.Watermark(rows, Cc, W, 3, 40.0, nslots=1, max_frames=F)
    xd = dev(torch, xs)
    ys, as_ = eng.makeWatermark(xd, xd, mk)
    cs = eng.detectWatermark(ys, mk)
    yh = ys.cpu().numpy()
    for f in range(F):
        so, yo, ao = O.embed(xs[f], xs[f], W, mask=omk)
        assert as_[f] == pytest.approx(ao, rel=TOL_A), f
        if f in (0, F - 1):
            np.testing.assert_allclose(yh[f], yo, rtol=0, atol=TOL_Y)
        assert cs[f] == pytest.approx(O.detect(yh[f], W, mask=omk)[1], abs=TOL_CORR), f
    eng.close()


@pytest.mark.parametrize("cols", [4, 8, 244, 248, 252, 256, 260, 492, 496, 500, 504, 508, 512, 740, 744, 748, 992, 1000])
@pytest.mark.parametrize("dtype", ["f32", "u8"])
def test_detect_overlapped_strips_at_every_boundary_case(wm, torch_cuda, cols, dtype):
    """k_detect's aligned 3x3 path works on strips 248 columns apart that load 256 (lanes 0 / 63 only provide neighbours): widths
    around every multiple of 248 and 256 put the image's last column in an owned lane, in the last owned lane, in a provider
    lane's chunk, and make the last strip own 1 lane or all 62.  Batched sweeps (a 5-frame batch: frame quads + a short last
    quad; and one frame: four segments per block), both masks, against the oracle and against the generic path (an unaligned
    view of the same frame): the two instantiations must agree to the grouping of the partial sums"""
    torch = torch_cuda
    R, F = 23, 5
    npdt = np.float32 if dtype == "f32" else np.uint8
    xs = np.stack([synth_frame(R, cols, frame=20 + f, dtype=npdt) for f in range(F)])
    W = synth_watermark(R, cols)
    eng = wm.Watermark(R, cols, W, 3, 40.0, nslots=1, max_frames=F)
    eng.set_fused(False)
    xd = dev(torch, xs)
    big = torch.zeros((F, R, cols + 8), dtype=xd.dtype, device="cuda")
    big[:, :, 1:cols + 1] = xd
    view = big[:, :, 1:cols + 1]                      # off the vector alignment: the generic instantiation
    for mk, omk in ((wm.MASK_TYPE.ME, O.MASK_ME), (wm.MASK_TYPE.NVF, O.MASK_NVF)):
        c_batch = eng.detectWatermark(xd, mk)
        c_gen = eng.detectWatermark(view, mk)
        c_one = eng.detectWatermark(xd[2], mk)
        for f in range(F):
            xf = xs[f].astype(np.float32)
            ref = O.detect(xf, W, mask=omk)[1] if dtype == "f32" else O.detect_u8(xs[f], W, mask=omk)[1]
            assert c_batch[f] == pytest.approx(ref, abs=TOL_CORR), (cols, f)
            assert c_gen[f] == pytest.approx(c_batch[f], abs=2e-6)
        assert c_one == pytest.approx(c_batch[2], abs=2e-6)
    eng.close()


@pytest.mark.parametrize("cols", [265, 266, 267, 497, 499, 501, 502, 503, 509, 510, 511, 513, 745, 747, 993, 1002, 1918])
@pytest.mark.parametrize("dtype", ["f32", "u8"])
def test_detect_widths_that_are_not_multiples_of_4(wm, torch_cuda, cols, dtype):
    """k_detect's 3x3 path on planes that allow vector access but whose width is not a multiple of 4 (the reference's 1918 x 1078
    sample, dense f32 planes of odd width, pitched FFmpeg frames): the overlapped strips own the columns below
    B = cols - cols % 4 - 4, ONE generic strip of 256 columns ending at the last column owns the rest (wm_march.hpp
    split_geom).  Widths around the multiples of 248 and 256 put B in the first, the last and the only owned lane of a strip, and
    the generic strip's first owned column anywhere in it.  Batches (frame quads + a short last quad) and single frames, both
    masks, against the oracle; u8 also against the all-generic path (a view off the dword alignment)."""
    torch = torch_cuda
    R, F = 23, 5
    npdt = np.float32 if dtype == "f32" else np.uint8
    xs = np.stack([synth_frame(R, cols, frame=40 + f, dtype=npdt) for f in range(F)])
    W = synth_watermark(R, cols)
    eng = wm.Watermark(R, cols, W, 3, 40.0, nslots=1, max_frames=F)
    eng.set_fused(False)
    pitch = (cols + 3) // 4 * 4 + 8
    big = torch.zeros((F, R, pitch), dtype=torch.float32 if dtype == "f32" else torch.uint8, device="cuda")
    big[:, :, :cols] = dev(torch, xs)
    pitched = big[:, :, :cols]                         # aligned base, pitch a multiple of 4, odd width: the split path (u8 and f32)
    dense = dev(torch, xs)                             # dense rows: the split path for f32 (4-byte aligned rows suffice)
    big2 = torch.zeros((F, R, pitch), dtype=big.dtype, device="cuda")
    big2[:, :, 1:cols + 1] = dense
    off1 = big2[:, :, 1:cols + 1]                      # u8: off the dword alignment -> the generic instance for the whole image
    for mk, omk in ((wm.MASK_TYPE.ME, O.MASK_ME), (wm.MASK_TYPE.NVF, O.MASK_NVF)):
        c_pitched = eng.detectWatermark(pitched, mk)
        c_dense = eng.detectWatermark(dense, mk)
        c_off = eng.detectWatermark(off1, mk)
        c_one = eng.detectWatermark(pitched[3], mk)
        for f in range(F):
            ref = O.detect(xs[f].astype(np.float32), W, mask=omk)[1] if dtype == "f32" else O.detect_u8(xs[f], W, mask=omk)[1]
            assert c_pitched[f] == pytest.approx(ref, abs=TOL_CORR), (cols, f)
            assert c_dense[f] == pytest.approx(c_pitched[f], abs=2e-6)
            assert c_off[f] == pytest.approx(c_pitched[f], abs=2e-6)
        assert c_one == pytest.approx(c_pitched[3], abs=2e-6)
    eng.close()
