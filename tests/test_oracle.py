"""CPU tests of the oracle: against the committed golden vectors, against the independent numpy
restatement, and the invariants SURVEY.md section 4 lists (PSNR, corr(marked) >> corr(unmarked),
passthrough on a constant image, determinism)."""
import numpy as np
import pytest

import np_restatement as NP
import oracle_lib as O
from synth import synth_frame, synth_watermark


def test_strength_factor():
    # Watermark.cpp:22 at psnr 40: 255/sqrt(10^4) = 2.55
    assert abs(O.strength_factor(40.0) - 2.55) < 1e-6
    assert abs(O.strength_factor(40.0) - float(NP.strength_factor(40.0))) < 1e-6


@pytest.mark.parametrize("tag", ["512", "720p_crop"])
def test_golden_scalars(golden, tag):
    from conftest import load_pair
    rgb, W = load_pair(golden, tag)
    gray = O.rgb2gray(rgb)
    g = golden[tag]
    st, c, e, m, mx = O.me_mask(gray)
    assert st == 0
    np.testing.assert_allclose(c, g["coefficients"], atol=1e-7)
    assert abs(mx - g["max_abs_e"]) < 1e-5
    for mask, name in ((O.MASK_ME, "ME"), (O.MASK_NVF, "NVF")):
        st, y, a = O.embed(gray, gray, W, mask=mask)
        assert st == 0 and a == pytest.approx(g[name]["a"], rel=1e-7)
        st, corr = O.detect(y, W, mask=mask)
        assert corr == pytest.approx(g[name]["corr_gray"], abs=1e-7)
        st, yrgb, a2 = O.embed(gray, rgb, W, mask=mask)
        assert a2 == a
        st, corr = O.detect(O.rgb2gray(yrgb), W, mask=mask)
        assert corr == pytest.approx(g[name]["corr_rgb_harness"], abs=1e-7)
        st, corr0 = O.detect(gray, W, mask=mask)
        assert corr0 == pytest.approx(g[name]["corr_unmarked"], abs=1e-7)
        assert abs(corr0) < 0.1 < corr


def test_golden_512_survey_values(golden):
    """SURVEY.md section 8c provisional values (independent numpy probe of the survey session)"""
    g = golden["512"]
    np.testing.assert_allclose(g["coefficients"], [-0.1694238, 0.4772605, -0.1014391, 0.2937111, 0.2937530,
                                                   -0.1014981, 0.4773920, -0.1693899], atol=2e-7)
    assert g["ME"]["a"] == pytest.approx(34.902996, rel=1e-6)
    assert g["ME"]["corr_rgb_harness"] == pytest.approx(0.7375435, abs=1e-6)
    assert g["NVF"]["a"] == pytest.approx(2.852794, rel=1e-6)
    assert g["NVF"]["corr_rgb_harness"] == pytest.approx(0.5858539, abs=1e-6)
    assert g["max_abs_e"] == pytest.approx(57.679, abs=1e-3)


def test_golden_crop_per_pixel(golden, pair_crop):
    import os
    from conftest import GOLDEN
    rgb, W = pair_crop
    gray = O.rgb2gray(rgb)
    exp = np.load(os.path.join(GOLDEN, "720p_crop_expected.npz"))
    np.testing.assert_array_equal(gray, exp["gray"])
    st, c, e, m, mx = O.me_mask(gray)
    np.testing.assert_array_equal(e, exp["e"])
    np.testing.assert_array_equal(m, exp["m_me"])
    np.testing.assert_array_equal(O.nvf_mask(gray), exp["m_nvf"])
    np.testing.assert_array_equal(O.embed(gray, gray, W, mask=O.MASK_ME)[1], exp["y_me"])
    np.testing.assert_array_equal(O.embed(gray, gray, W, mask=O.MASK_NVF)[1], exp["y_nvf"])


@pytest.mark.parametrize("shape", [(64, 64), (70, 131), (96, 200)])
def test_oracle_vs_numpy_restatement(shape):
    x = synth_frame(shape[0], shape[1], frame=3)
    W = synth_watermark(shape[0], shape[1])
    Rx, rx = O.gram(x)
    Rn, rn = NP.gram(x)
    np.testing.assert_allclose(Rx, Rn, rtol=1e-12)
    np.testing.assert_allclose(rx, rn, rtol=1e-12)
    st, c, e, m, mx = O.me_mask(x)
    cn = NP.coefficients(x)
    np.testing.assert_allclose(c, cn, atol=2e-6)
    np.testing.assert_allclose(O.scaled_neighbors(x, c), NP.scaled_neighbors(x, c), rtol=0, atol=1e-4)
    for p in (3, 5, 7, 9):
        np.testing.assert_allclose(O.nvf_mask(x, p), NP.nvf_mask(x, p), rtol=0, atol=1e-6)
    for mask, name in ((O.MASK_ME, "ME"), (O.MASK_NVF, "NVF")):
        st, y, a = O.embed(x, x, W, mask=mask)
        yn, an = NP.embed(x, x, W, mask=name)
        assert a == pytest.approx(an, rel=1e-5)
        np.testing.assert_allclose(y, yn, atol=2e-3)
        st, corr = O.detect(y, W, mask=mask)
        assert corr == pytest.approx(NP.detect(yn, W, mask=name), abs=1e-5)


@pytest.mark.parametrize("mask", [O.MASK_ME, O.MASK_NVF])
def test_psnr_and_detection_invariants(mask):
    x = synth_frame(128, 192, frame=0)
    W = synth_watermark(128, 192)
    for psnr in (35.0, 40.0, 45.0):
        st, y_unclamped_proxy, a = O.embed(x, x, W, psnr=psnr, mask=mask)
        mse = np.mean((y_unclamped_proxy.astype(np.float64) - x) ** 2)
        got = 10 * np.log10(255.0 ** 2 / mse)
        assert abs(got - psnr) < 0.1  # clamping only lowers the distortion slightly
    st, y, a = O.embed(x, x, W, mask=mask)
    st, c1 = O.detect(y, W, mask=mask)
    st, c0 = O.detect(x, W, mask=mask)
    assert c1 > 0.3 and abs(c0) < 0.05


def test_unsolvable_passthrough():
    """constant image => rank-1 Rx => embed returns base unmodified, detect returns 0 (Watermark.cpp:164-165,246-247)"""
    x = np.full((64, 80), 117.0, np.float32)
    W = synth_watermark(64, 80)
    base = synth_frame(64, 80, frame=1)
    st, y, a = O.embed(x, base, W, mask=O.MASK_ME)
    assert st == O.UNSOLVABLE and np.isnan(a)
    np.testing.assert_array_equal(y, base)
    st, corr = O.detect(x, W, mask=O.MASK_ME)
    assert st == O.UNSOLVABLE and corr == 0.0
    st, corr = O.detect(x, W, mask=O.MASK_NVF)
    assert st == O.UNSOLVABLE and corr == 0.0
    st, sol = O.solve(np.zeros((8, 8)), np.zeros(8))
    assert st == O.UNSOLVABLE


def test_bad_p():
    x = synth_frame(64, 64)
    with pytest.raises(ValueError):
        O.nvf_mask(x, p=4)
    st, y, a = O.embed(x, x, synth_watermark(64, 64), p=5, mask=O.MASK_ME)  # ME only for p=3 (main.cpp:89)
    assert st < 0


def test_reference_noise_switches():
    """fp16 products / f32 sums (the reference's own arithmetic, me_p3.hpp:8-21,65-66) move the
    coefficients only within the SURVEY section 8c bracket"""
    x = synth_frame(256, 256, frame=2)
    st, c, *_ = O.me_mask(x)
    st, c16, *_ = O.me_mask(x, fp16_products=True, accum_f32=True)
    assert 0 < np.abs(c - c16).max() < 5e-3


def test_video_u8_contract(golden, pair512):
    rgb, W = pair512
    g8 = O.rgb2gray(rgb).astype(np.uint8)
    st, y8, a = O.embed_u8(g8, W)
    assert a == pytest.approx(golden["512"]["ME"]["video_u8"]["a"], rel=1e-7)
    st, corr = O.detect_u8(y8, W)
    assert corr == pytest.approx(golden["512"]["ME"]["video_u8"]["corr"], abs=1e-7)
    assert y8.dtype == np.uint8 and int(np.abs(y8.astype(int) - g8).max()) > 0


def test_determinism():
    x = synth_frame(100, 140, frame=5)
    W = synth_watermark(100, 140)
    r1 = O.embed(x, x, W)
    r2 = O.embed(x, x, W)
    np.testing.assert_array_equal(r1[1], r2[1])
    assert r1[2] == r2[2]
