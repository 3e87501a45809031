"""BASELINE.json configurations at full size on the GPU: against the oracle where it finishes in seconds, and through
size-independent properties (achieved PSNR, marked >> unmarked correlation, linearity of the embedded signal in
the strength factor, round trip embed -> detect, run-to-run determinism)."""
import numpy as np
import pytest

import oracle_lib as O
from synth import synth_watermark

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["fused", "sweeps"])
def single_call_path(request, monkeypatch):
    """every test runs twice: synchronous one-frame calls on the fused single-launch kernels (the default) and on the
    batched sweeps (WM_FUSED=0, read when an engine is created)"""
    monkeypatch.setenv("WM_FUSED", "1" if request.param == "fused" else "0")
    return request.param


@pytest.fixture(scope="module")
def tc():
    import torch
    assert torch.cuda.is_available()
    return torch


def frames_gpu(wm, tc, R, C, n, dtype="f32", first=0):
    import importlib
    synth = importlib.import_module("watermarking-gpu_amd.synth")
    return synth.synth_frames_torch(R, C, n, "cuda", dtype=dtype, first_frame=first)


def psnr(a, b):
    mse = float(((a.double() - b.double()) ** 2).mean())
    return 10 * np.log10(255.0 ** 2 / mse)


def test_config1_1080p_nvf_and_me_vs_oracle(wm, tc):
    """BASELINE configs[1]: 1920x1080 single image, NVF + ME masks, embed+detect"""
    R, C = 1080, 1920
    W = synth_watermark(R, C)
    x = frames_gpu(wm, tc, R, C, 1)[0]
    xh = x.cpu().numpy()
    eng = wm.Watermark(R, C, W, 3, 40.0)
    for mt in (wm.MASK_TYPE.NVF, wm.MASK_TYPE.ME):
        y, a = eng.makeWatermark(x, x, mt)
        st, yo, ao = O.embed(xh, xh, W, mask=int(mt))
        assert a == pytest.approx(ao, rel=1e-4)
        np.testing.assert_allclose(y.cpu().numpy(), yo, rtol=0, atol=1e-3)
        corr = eng.detectWatermark(y, mt)
        assert corr == pytest.approx(O.detect(yo, W, mask=int(mt))[1], abs=1e-5)
        assert abs(psnr(y, x) - 40.0) < 0.1


def test_config2_4k_me_vs_oracle_and_properties(wm, tc):
    """BASELINE configs[2]: 3840x2160, ME mask (the roofline run's workload), one frame against the oracle"""
    R, C = 2160, 3840
    W = synth_watermark(R, C)
    xs = frames_gpu(wm, tc, R, C, 3)
    eng = wm.Watermark(R, C, W, 3, 40.0, nslots=2, max_frames=3)
    ys, a = eng.makeWatermark(xs, xs, wm.MASK_TYPE.ME)
    corr = eng.detectWatermark(ys, wm.MASK_TYPE.ME)
    corr0 = eng.detectWatermark(xs, wm.MASK_TYPE.ME)
    xh = xs[0].cpu().numpy()
    st, yo, ao = O.embed(xh, xh, W)
    assert a[0] == pytest.approx(ao, rel=1e-4)
    np.testing.assert_allclose(ys[0].cpu().numpy(), yo, rtol=0, atol=1e-3)
    assert corr[0] == pytest.approx(O.detect(yo, W)[1], abs=1e-5)
    Rx, rx = eng.gram(xs[0])
    Ro, ro = O.gram(xh)
    np.testing.assert_allclose(Rx, Ro, rtol=1e-13)
    for f in range(3):
        assert abs(psnr(ys[f], xs[f]) - 40.0) < 0.1          # embed hits the requested PSNR
        assert corr[f] > 0.3 and abs(corr0[f]) < 0.01          # marked >> unmarked ~ 0
    # determinism: same batch again, bitwise
    ys2, a2 = eng.makeWatermark(xs, xs, wm.MASK_TYPE.ME)
    assert a2 == a and tc.equal(ys, ys2)
    # linearity in the strength factor: y - x scales as 10^(-dPSNR/20) while unclamped
    eng2 = wm.Watermark(R, C, W, 3, 46.0206)  # half the amplitude
    yh, ah = eng2.makeWatermark(xs[0], xs[0], wm.MASK_TYPE.ME)
    assert ah == pytest.approx(a[0] / 2.0, rel=1e-4)
    d1 = (ys[0] - xs[0]).double()
    d2 = (yh - xs[0]).double()
    interior = (xs[0] > 30) & (xs[0] < 225)
    assert float((d1[interior] - 2.0 * d2[interior]).abs().max()) < 2e-3


def test_config4_8k_nvf_and_me(wm, tc):
    """BASELINE configs[4]: 7680x4320, NVF + ME (the strip march keeps LDS/registers independent of the image size).
    Oracle comparison on the strength and the correlation; per-pixel comparison on a band of rows."""
    R, C = 4320, 7680
    W = synth_watermark(R, C)
    x = frames_gpu(wm, tc, R, C, 1)[0]
    xh = x.cpu().numpy()
    eng = wm.Watermark(R, C, W, 3, 40.0)
    for mt in (wm.MASK_TYPE.ME, wm.MASK_TYPE.NVF):
        y, a = eng.makeWatermark(x, x, mt)
        corr = eng.detectWatermark(y, mt)
        st, yo, ao = O.embed(xh, xh, W, mask=int(mt))
        assert a == pytest.approx(ao, rel=1e-4)
        np.testing.assert_allclose(y[2000:2300].cpu().numpy(), yo[2000:2300], rtol=0, atol=1e-3)
        np.testing.assert_allclose(y[:8].cpu().numpy(), yo[:8], rtol=0, atol=1e-3)
        np.testing.assert_allclose(y[-8:].cpu().numpy(), yo[-8:], rtol=0, atol=1e-3)
        assert corr == pytest.approx(O.detect(yo, W, mask=int(mt))[1], abs=1e-5)
        assert abs(psnr(y, x) - 40.0) < 0.1


@pytest.mark.parametrize("shape", [(1078, 1918), (2160, 3872)])
def test_reference_odd_sample_shapes(wm, tc, shape):
    """the reference's odd-shaped samples (make_w.bat: 1918x1078 not a multiple of 16; 4k_non_divisible 3872x2160 not
    a multiple of 64/256): ragged last strip + unaligned pitch take the LDS path"""
    R, C = shape
    W = synth_watermark(R, C)
    x = frames_gpu(wm, tc, R, C, 1)[0]
    xh = x.cpu().numpy()
    eng = wm.Watermark(R, C, W, 3, 40.0)
    y, a = eng.makeWatermark(x, x, wm.MASK_TYPE.ME)
    st, yo, ao = O.embed(xh, xh, W)
    assert a == pytest.approx(ao, rel=1e-4)
    np.testing.assert_allclose(y.cpu().numpy(), yo, rtol=0, atol=1e-3)
    assert eng.detectWatermark(y, wm.MASK_TYPE.ME) == pytest.approx(O.detect(yo, W)[1], abs=1e-5)
    m, _, _, _ = eng.computeMask(x, wm.MASK_TYPE.NVF)
    np.testing.assert_array_equal(m.cpu().numpy(), O.nvf_mask(xh))


def test_video_stream_u8_4k_batch(wm, tc):
    """BASELINE configs[3] shape: 3840x2160 u8 Y planes, watermark every frame, a batch per launch"""
    R, C, F = 2160, 3840, 4
    W = synth_watermark(R, C)
    xs = frames_gpu(wm, tc, R, C, F, dtype="u8")
    eng = wm.Watermark(R, C, W, 3, 40.0, nslots=2, max_frames=F)
    ys, a = eng.makeWatermark(xs, xs, wm.MASK_TYPE.ME)
    corr = eng.detectWatermark(ys, wm.MASK_TYPE.ME)
    st, yo, ao = O.embed_u8(xs[1].cpu().numpy(), W)
    assert a[1] == pytest.approx(ao, rel=1e-4)
    d = np.abs(ys[1].cpu().numpy().astype(int) - yo.astype(int))
    assert d.max() <= 1 and (d != 0).mean() <= 1e-3
    assert corr[1] == pytest.approx(O.detect_u8(ys[1].cpu().numpy(), W)[1], abs=1e-5)
    assert all(c > 0.3 for c in corr)
