"""GPU (through the C ABI) against the oracle in the REFERENCE's arithmetic (ref_arith: half products, 64-lane f32
work-group sums, f32 fold, f32 LU) on the committed fixtures and on synthetic frames at the BASELINE sizes: the bracket
SURVEY.md 8c states for "build vs the reference OpenCL path".  See tests/test_reference_bracket.py for what this can and
cannot show (parity stays unpinned: the reference ships no vectors)."""
import numpy as np
import pytest

import oracle_lib as O
from synth import synth_frame, synth_watermark
from test_reference_bracket import BRACKET

pytestmark = pytest.mark.gpu


def check(wm, torch, x, W, bracket, psnr_req=40.0):
    tc, ta, tcorr, trms = bracket
    R, Cc = x.shape
    eng = wm.Watermark(R, Cc, W, 3, psnr_req)
    xd = torch.from_numpy(np.ascontiguousarray(x)).cuda()
    m, e, c, st = eng.computeMask(xd, wm.MASK_TYPE.ME)
    st_r, cr, er, mr, mxr = O.me_mask(x, ref_arith=True)
    assert st == 0 and st_r == 0
    assert np.abs(c - cr).max() <= tc, ("coefficients", np.abs(c - cr).max())
    y, a = eng.makeWatermark(xd, xd, wm.MASK_TYPE.ME)
    so, yr, ar = O.embed(x, x, W, ref_arith=True)
    assert abs(a - ar) <= ta * abs(ar), ("a", a, ar)
    yh = y.cpu().numpy().astype(np.float64)
    assert np.sqrt(((yh - yr) ** 2).mean()) <= trms
    psnr_gpu = 10 * np.log10(255.0 ** 2 / ((yh - x) ** 2).mean())
    psnr_ref = 10 * np.log10(255.0 ** 2 / ((yr.astype(np.float64) - x) ** 2).mean())
    assert abs(psnr_gpu - psnr_ref) <= 0.05 and psnr_gpu >= psnr_req - 0.05, (psnr_gpu, psnr_ref)
    corr = eng.detectWatermark(y, wm.MASK_TYPE.ME)
    corr_r = O.detect(y.cpu().numpy(), W, ref_arith=True)[1]
    assert abs(corr - corr_r) <= tcorr, ("corr", corr, corr_r)
    # NVF needs no prediction system for the mask, but its detector does (e_w, e_u use the coefficients)
    yn, an = eng.makeWatermark(xd, xd, wm.MASK_TYPE.NVF)
    cn = eng.detectWatermark(yn, wm.MASK_TYPE.NVF)
    assert abs(cn - O.detect(yn.cpu().numpy(), W, mask=O.MASK_NVF, ref_arith=True)[1]) <= tcorr
    eng.close()


@pytest.mark.parametrize("tag", ["512", "720p_crop"])
def test_gpu_vs_reference_arithmetic_fixtures(wm, golden, tag, pair512, pair_crop):
    import torch
    rgb, W = pair512 if tag == "512" else pair_crop
    check(wm, torch, O.rgb2gray(rgb), W, BRACKET[tag])


@pytest.mark.parametrize("shape,bracket", [((1080, 1920), (5e-3, 1e-2, 2e-3, 0.25)), ((2160, 3840), (5e-2, 1e-2, 2e-3, 0.25))])
def test_gpu_vs_reference_arithmetic_baseline_sizes(wm, shape, bracket):
    import torch
    R, Cc = shape
    check(wm, torch, synth_frame(R, Cc, frame=0), synth_watermark(R, Cc), bracket)
