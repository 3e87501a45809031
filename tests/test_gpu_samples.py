"""The reference's own LARGE sample images on the HIP path (samples/images/{1080p,4k,4k_non_divisible}.png; grey u8 planes
committed under tests/golden/ by make_golden.py): real content, cond(Rx) = 1e4 at 1080p and 1.4e5 at 4K -- the regime the exact
f64 Gram exists for -- against the golden scalars (produced by the oracle in the build container) and against the oracle run
here, on the fused single-launch kernels and on the batched sweeps (3872x2160 takes the sweeps only: 16 strips x 17 bands do
not fit one workgroup per CU), as f32 images and as u8 video planes, both masks.  The reference ships no W for these images
(make_w.bat covers 512/480p/720p): W is this build's generator at the reference's seed (wm_genw rows cols 28390211).

The golden scalars cannot pin parity (the same restatement produced them); what this adds is the reference's hardest inputs
on the GPU path, and the reference-arithmetic bracket on them."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O
from synth import synth_watermark

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
TOL_A, TOL_CORR, TOL_Y = 1e-4, 1e-5, 1e-3
SAMPLES = ["1080p", "4k", "4k_non_divisible"]


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(GOLD, "golden.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def samples():
    cache = {}

    def get(tag):
        if tag not in cache:
            g8 = np.load(os.path.join(GOLD, f"{tag}_gray_u8.npz"))["gray"]
            cache[tag] = (g8, synth_watermark(*g8.shape))
        return cache[tag]
    return get


def dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("tag", SAMPLES)
@pytest.mark.parametrize("path", ["fused", "sweeps"])
def test_reference_sample_f32(wm, golden, samples, tag, path):
    """makeWatermark / detectWatermark on the sample as an f32 image, ME and NVF: strength, output plane and correlation against
    the golden record and the live oracle; the fused kernels must really have run where the shape allows them"""
    import torch
    g8, W = samples(tag)
    R, Cc = g8.shape
    rec = golden[tag]
    x = g8.astype(np.float32)
    eng = wm.Watermark(R, Cc, W, 3, 40.0)
    eng.set_fused(path == "fused")
    fusable = (R, Cc) != (2160, 3872)
    assert eng.fused_info()[0] == (path == "fused" and fusable)
    xd = dev(torch, x)
    m, e, c, st = eng.computeMask(xd, wm.MASK_TYPE.ME, want_error_sequence=True)
    assert st == 0
    np.testing.assert_allclose(c, np.array(rec["coefficients"], np.float32), rtol=0, atol=2e-6)   # cond 1.4e5: a few f32 ulps of O(1) values
    for mk, omk, name in ((wm.MASK_TYPE.ME, O.MASK_ME, "ME"), (wm.MASK_TYPE.NVF, O.MASK_NVF, "NVF")):
        y, a = eng.makeWatermark(xd, xd, mk)
        assert a == pytest.approx(rec[name]["a"], rel=TOL_A)
        so, yo, ao = O.embed(x, x, W, mask=omk)
        assert so == 0 and ao == pytest.approx(rec[name]["a"], rel=1e-6)
        np.testing.assert_allclose(y.cpu().numpy(), yo, rtol=0, atol=TOL_Y)
        corr = eng.detectWatermark(dev(torch, yo), mk)
        assert corr == pytest.approx(rec[name]["corr_gray"], abs=TOL_CORR)
        assert eng.detectWatermark(y, mk) == pytest.approx(rec[name]["corr_gray"], abs=2e-5)       # on the GPU's own output
        assert eng.detectWatermark(xd, mk) == pytest.approx(rec[name]["corr_unmarked"], abs=TOL_CORR)
    assert eng.fused_info()[3] == 0
    eng.close()


@pytest.mark.parametrize("tag", SAMPLES)
@pytest.mark.parametrize("path", ["fused", "sweeps"])
def test_reference_sample_u8_video_plane(wm, golden, samples, tag, path):
    """the same planes through the video flow (u8 in, u8 out by truncation, in place: main.cpp:355-357)"""
    import torch
    g8, W = samples(tag)
    R, Cc = g8.shape
    rec = golden[tag]
    eng = wm.Watermark(R, Cc, W, 3, 40.0)
    eng.set_fused(path == "fused")
    for mk, omk, name in ((wm.MASK_TYPE.ME, O.MASK_ME, "ME"), (wm.MASK_TYPE.NVF, O.MASK_NVF, "NVF")):
        so, yo, ao = O.embed_u8(g8, W, mask=omk)
        frame = dev(torch, g8)
        y, a = eng.makeWatermark(frame, frame, mk, out=frame)
        assert a == pytest.approx(rec[name]["video_u8"]["a"], rel=TOL_A)
        d = np.abs(frame.cpu().numpy().astype(int) - yo.astype(int))
        assert d.max() <= 1 and (d != 0).mean() <= 1e-3
        assert eng.detectWatermark(dev(torch, yo), mk) == pytest.approx(rec[name]["video_u8"]["corr"], abs=TOL_CORR)
    assert eng.fused_info()[3] == 0
    eng.close()


@pytest.mark.parametrize("tag", SAMPLES)
def test_reference_sample_batched_sweeps_and_reference_bracket(wm, golden, samples, tag):
    """the sample inside a batch of 4 (the batched launches: frame quads, shared W tiles) gives the single-call results; and
    the GPU's results sit inside the bracket SURVEY.md 8c states around the reference's own arithmetic (half products, 64-lane
    f32 sums, f32 LU -- golden.json `reference_arith`) on these ill-conditioned frames"""
    import torch
    g8, W = samples(tag)
    R, Cc = g8.shape
    rec = golden[tag]
    x = g8.astype(np.float32)
    F = 4
    xs = np.stack([x, x[::-1].copy(), x, x[:, ::-1].copy()])
    eng = wm.Watermark(R, Cc, W, 3, 40.0, nslots=1, max_frames=F)
    ys, a = eng.makeWatermark(dev(torch, xs), dev(torch, xs), wm.MASK_TYPE.ME)
    cs = eng.detectWatermark(ys, wm.MASK_TYPE.ME)
    assert a[0] == a[2] and cs[0] == cs[2] and torch.equal(ys[0], ys[2])
    assert a[0] == pytest.approx(rec["ME"]["a"], rel=TOL_A) and cs[0] == pytest.approx(rec["ME"]["corr_gray"], abs=2e-5)
    ra = rec["reference_arith"]
    assert abs(a[0] - ra["a_ME"]) <= 1e-2 * abs(ra["a_ME"])
    assert abs(cs[0] - ra["corr_ME_on_exact_y"]) <= 2e-3
    assert ra["max_abs_dcoef"] <= 5e-2 and ra["y_rms_vs_exact"] <= 0.25
    eng.close()
