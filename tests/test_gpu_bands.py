"""Row-band intra-frame sharding (include/wm.h wm_band_*; watermarking-gpu_amd/bands.py).

(1) building blocks in one process: the bands' partial totals add up to the whole image's, the stitched band outputs
    equal the whole-image embed, the summed detector sums give the whole-image correlation;
(2) the torch.distributed orchestration with two ranks on the one GPU of the box (gloo carries the exchange)."""
import importlib
import os
import sys

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


def corr_of(d, nu, nw):
    return float(np.float32(d) / np.float32(np.sqrt(nw) * np.sqrt(nu)))


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("shape", [(120, 300), (257, 764), (96, 1030)])
@pytest.mark.parametrize("mask", ["ME", "NVF"])
def test_band_building_blocks(wm, tc, world, shape, mask):
    torch = tc
    bands = importlib.import_module("watermarking-gpu_amd.bands")
    R, Cc = shape
    x = synth_frame(R, Cc, frame=3)
    W = synth_watermark(R, Cc)
    mk = wm.MASK_TYPE[mask]
    omk = O.MASK_ME if mask == "ME" else O.MASK_NVF
    full = wm.Watermark(R, Cc, W, 3, 40.0)
    xd = torch.from_numpy(x).cuda()
    y_full, a_full = full.makeWatermark(xd, xd, mk)
    c_full = full.detectWatermark(y_full, mk)
    tot_full = full.gram_totals(xd)

    engs, views = [], []
    for r in range(world):
        g0, g1, lo, hi = bands.band_with_halo(R, r, world)
        e = wm.Watermark(g1 - g0, Cc, np.ascontiguousarray(W[g0:g1]), 3, 40.0)
        e.band_configure(lo, hi, R)
        engs.append((e, g0, g1, lo, hi))
        views.append(xd[g0:g1].contiguous())
    # Gram: exact sums -> the bands' totals add up to the whole image's
    tot = sum(e.gram_totals(v) for (e, *_), v in zip(engs, views))
    np.testing.assert_allclose(tot, tot_full, rtol=1e-13)
    # embed
    if mask == "ME":
        for (e, *_), v in zip(engs, views):
            assert e.band_solve(tot) == 0
    st = [e.band_stats(v, mk) for (e, *_), v in zip(engs, views)]
    mx, ss = max(s[0] for s in st), sum(s[1] for s in st)
    y = torch.empty_like(xd)
    a = None
    for (e, g0, g1, lo, hi), v in zip(engs, views):
        out = v.clone()
        a = e.band_embed(v, v, out, mk, mx, ss)
        y[g0 + lo:g0 + hi] = out[lo:hi]
        # rows outside the owned range are untouched
        assert torch.equal(out[:lo], v[:lo]) and torch.equal(out[hi:], v[hi:])
    assert a == pytest.approx(a_full, rel=1e-6)
    np.testing.assert_allclose(y.cpu().numpy(), y_full.cpu().numpy(), rtol=0, atol=1e-4)
    so, yo, ao = O.embed(x, x, W, mask=omk)
    assert a == pytest.approx(ao, rel=1e-4)
    np.testing.assert_allclose(y.cpu().numpy(), yo, rtol=0, atol=1e-3)
    # detect on the stitched image (halo rows = neighbours' owned rows)
    toty = sum(e.gram_totals(y[g0:g1].contiguous()) for (e, g0, g1, lo, hi) in engs)
    sums = np.zeros(3)
    for (e, g0, g1, lo, hi) in engs:
        assert e.band_solve(toty) == 0
        sums += np.array(e.band_detect_sums(y[g0:g1].contiguous(), mk))
    c = corr_of(*sums)
    assert c == pytest.approx(full.detectWatermark(y, mk), abs=2e-6)
    assert c == pytest.approx(O.detect(y.cpu().numpy(), W, mask=omk)[1], abs=1e-5)
    assert c == pytest.approx(c_full, abs=1e-4)
    for e, *_ in engs:
        e.close()
    full.close()


@pytest.mark.parametrize("p", [5, 7, 9])
def test_band_nvf_larger_windows(wm, tc, p):
    """NVF with p > 3 in row bands: the halo is p//2 + 1 rows (the detector reads the mask one row away from the pixel
    it scores, the mask reads x p//2 rows further); stitched embed and summed detector sums against the whole image"""
    torch = tc
    bands = importlib.import_module("watermarking-gpu_amd.bands")
    R, Cc, world = 150, 516, 3
    x = synth_frame(R, Cc, frame=5)
    W = synth_watermark(R, Cc)
    mk = wm.MASK_TYPE.NVF
    full = wm.Watermark(R, Cc, W, p, 40.0)
    xd = torch.from_numpy(x).cuda()
    y_full, a_full = full.makeWatermark(xd, xd, mk)
    c_full = full.detectWatermark(y_full, mk)
    halo = bands.halo_rows(p)
    assert halo == p // 2 + 1
    engs = []
    for r in range(world):
        g0, g1, lo, hi = bands.band_with_halo(R, r, world, halo)
        e = wm.Watermark(g1 - g0, Cc, np.ascontiguousarray(W[g0:g1]), p, 40.0)
        if lo > 0:
            with pytest.raises(RuntimeError, match="halo rows"):
                e.band_configure(2, hi, R)  # the 2 rows that suffice for p = 3 do not for p >= 5
        e.band_configure(lo, hi, R)
        engs.append((e, g0, g1, lo, hi))
    st = [e.band_stats(xd[g0:g1].contiguous(), mk) for (e, g0, g1, lo, hi) in engs]
    mx, ss = max(s[0] for s in st), sum(s[1] for s in st)
    y = torch.empty_like(xd)
    for (e, g0, g1, lo, hi) in engs:
        v = xd[g0:g1].contiguous()
        out = v.clone()
        a = e.band_embed(v, v, out, mk, mx, ss)
        y[g0 + lo:g0 + hi] = out[lo:hi]
    assert a == pytest.approx(a_full, rel=1e-6)
    np.testing.assert_allclose(y.cpu().numpy(), y_full.cpu().numpy(), rtol=0, atol=1e-4)
    toty = sum(e.gram_totals(y[g0:g1].contiguous()) for (e, g0, g1, lo, hi) in engs)
    sums = np.zeros(3)
    for (e, g0, g1, lo, hi) in engs:
        assert e.band_solve(toty) == 0
        sums += np.array(e.band_detect_sums(y[g0:g1].contiguous(), mk))
    assert corr_of(*sums) == pytest.approx(c_full, abs=2e-6)
    assert corr_of(*sums) == pytest.approx(O.detect(y.cpu().numpy(), W, mask=O.MASK_NVF, p=p)[1], abs=1e-5)
    for e, *_ in engs:
        e.close()
    full.close()


def test_band_argument_errors(wm, tc):
    W = synth_watermark(40, 64)
    eng = wm.Watermark(40, 64, W, 3, 40.0)
    with pytest.raises(RuntimeError):
        eng.band_configure(1, 30, 100)     # an interior top side needs 2 halo rows
    with pytest.raises(RuntimeError):
        eng.band_configure(2, 39, 100)     # ... and so does an interior bottom side
    with pytest.raises(RuntimeError):
        eng.band_configure(10, 5, 100)
    eng.band_configure(2, 38, 100)
    eng.band_configure(0, 0, 0)            # off again
    x = tc.from_numpy(synth_frame(40, 64)).cuda()
    y, a = eng.makeWatermark(x, x, wm.MASK_TYPE.ME)
    so, yo, ao = O.embed(synth_frame(40, 64), synth_frame(40, 64), W)
    assert a == pytest.approx(ao, rel=1e-4)
    eng.close()


def _rank_main(rank, world, port, R, Cc, out_dir):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    wm = importlib.import_module("watermarking-gpu_amd")
    bands = importlib.import_module("watermarking-gpu_amd.bands")
    from synth import synth_frame, synth_watermark
    x = synth_frame(R, Cc, frame=3)
    W = synth_watermark(R, Cc)
    bw = bands.BandedWatermark(R, Cc, W, 3, 40.0, rank, world, device=0, coll_device="cpu")
    band = torch.from_numpy(x[bw.g0:bw.g1]).cuda()
    res = {}
    for name in ("ME", "NVF"):
        mk = wm.MASK_TYPE[name]
        yb, a = bw.embed(band, mk)
        yb = bw.exchange_halos(yb)
        c = bw.detect(yb, mk)
        res[name] = (a, c, yb[bw.own_lo:bw.own_hi].cpu().numpy())
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), a_me=res["ME"][0], c_me=res["ME"][1], y_me=res["ME"][2],
             a_nvf=res["NVF"][0], c_nvf=res["NVF"][1], y_nvf=res["NVF"][2])
    bw.close()
    dist.destroy_process_group()


def test_banded_engine_two_ranks(wm, tc, tmp_path):
    """bands.BandedWatermark end to end: 2 processes, gloo for the exchange, both on the box's one GPU"""
    import torch.multiprocessing as mp
    R, Cc, world = 203, 508, 2
    port = 29000 + (os.getpid() % 1000)
    mp.spawn(_rank_main, args=(world, port, R, Cc, str(tmp_path)), nprocs=world, join=True)
    x = synth_frame(R, Cc, frame=3)
    W = synth_watermark(R, Cc)
    r = [np.load(tmp_path / f"rank{k}.npz") for k in range(world)]
    for tag, omk in (("me", O.MASK_ME), ("nvf", O.MASK_NVF)):
        y = np.concatenate([rk[f"y_{tag}"] for rk in r])
        so, yo, ao = O.embed(x, x, W, mask=omk)
        assert y.shape == yo.shape
        np.testing.assert_allclose(y, yo, rtol=0, atol=1e-3)
        assert float(r[0][f"a_{tag}"]) == float(r[1][f"a_{tag}"]) == pytest.approx(ao, rel=1e-4)
        assert float(r[0][f"c_{tag}"]) == float(r[1][f"c_{tag}"]) == pytest.approx(O.detect(yo, W, mask=omk)[1], abs=1e-5)


@pytest.mark.parametrize("world", [2, 4])
def test_band_building_blocks_u8(wm, tc, world):
    """u8 Y planes (the video contract) in row bands: integer Gram path clipped to the owned rows, u8 stores of the
    owned rows only"""
    torch = tc
    bands = importlib.import_module("watermarking-gpu_amd.bands")
    R, Cc = 230, 512
    x = synth_frame(R, Cc, frame=9, dtype=np.uint8)
    W = synth_watermark(R, Cc)
    mk = wm.MASK_TYPE.ME
    full = wm.Watermark(R, Cc, W, 3, 40.0)
    xd = torch.from_numpy(x).cuda()
    y_full, a_full = full.makeWatermark(xd, xd, mk)
    tot_full = full.gram_totals(xd)
    engs = []
    for r in range(world):
        g0, g1, lo, hi = bands.band_with_halo(R, r, world)
        e = wm.Watermark(g1 - g0, Cc, np.ascontiguousarray(W[g0:g1]), 3, 40.0)
        e.band_configure(lo, hi, R)
        engs.append((e, g0, g1, lo, hi))
    tot = sum(e.gram_totals(xd[g0:g1].contiguous()) for (e, g0, g1, lo, hi) in engs)
    np.testing.assert_array_equal(tot, tot_full)  # integer sums: exactly equal
    st = []
    for (e, g0, g1, lo, hi) in engs:
        assert e.band_solve(tot) == 0
        st.append(e.band_stats(xd[g0:g1].contiguous(), mk))
    mx, ss = max(s[0] for s in st), sum(s[1] for s in st)
    y = torch.empty_like(xd)
    for (e, g0, g1, lo, hi) in engs:
        v = xd[g0:g1].contiguous()
        out = v.clone()
        a = e.band_embed(v, v, out, mk, mx, ss)
        y[g0 + lo:g0 + hi] = out[lo:hi]
        assert torch.equal(out[:lo], v[:lo]) and torch.equal(out[hi:], v[hi:])
    assert a == pytest.approx(a_full, rel=1e-6)
    d = (y.to(torch.int32) - y_full.to(torch.int32)).abs()
    assert int(d.max()) <= 1 and float((d != 0).float().mean()) <= 1e-3
    toty = sum(e.gram_totals(y[g0:g1].contiguous()) for (e, g0, g1, lo, hi) in engs)
    sums = np.zeros(3)
    for (e, g0, g1, lo, hi) in engs:
        assert e.band_solve(toty) == 0
        sums += np.array(e.band_detect_sums(y[g0:g1].contiguous(), mk))
    assert corr_of(*sums) == pytest.approx(full.detectWatermark(y, mk), abs=2e-6)
    for e, *_ in engs:
        e.close()
    full.close()


def test_band_embed_rgb_base(wm, tc):
    """image mode in bands: grey mask source, planar RGB base (main.cpp:169-190) -- the owned rows of all three channels"""
    torch = tc
    bands = importlib.import_module("watermarking-gpu_amd.bands")
    R, Cc, world = 150, 320, 3
    x = synth_frame(R, Cc, frame=1)
    rgb = np.stack([np.clip(x + 12.0 * (k - 1), 0, 255).astype(np.float32) for k in range(3)])
    W = synth_watermark(R, Cc)
    mk = wm.MASK_TYPE.ME
    full = wm.Watermark(R, Cc, W, 3, 40.0)
    xd, rd = torch.from_numpy(x).cuda(), torch.from_numpy(rgb).cuda()
    y_full, a_full = full.makeWatermark(xd, rd, mk)
    engs = []
    for r in range(world):
        g0, g1, lo, hi = bands.band_with_halo(R, r, world)
        e = wm.Watermark(g1 - g0, Cc, np.ascontiguousarray(W[g0:g1]), 3, 40.0)
        e.band_configure(lo, hi, R)
        engs.append((e, g0, g1, lo, hi))
    tot = sum(e.gram_totals(xd[g0:g1].contiguous()) for (e, g0, g1, lo, hi) in engs)
    st = []
    for (e, g0, g1, lo, hi) in engs:
        assert e.band_solve(tot) == 0
        st.append(e.band_stats(xd[g0:g1].contiguous(), mk))
    mx, ss = max(s[0] for s in st), sum(s[1] for s in st)
    y = torch.empty_like(rd)
    for (e, g0, g1, lo, hi) in engs:
        v, b = xd[g0:g1].contiguous(), rd[:, g0:g1].contiguous()
        out = b.clone()
        a = e.band_embed(v, b, out, mk, mx, ss)
        y[:, g0 + lo:g0 + hi] = out[:, lo:hi]
        assert torch.equal(out[:, :lo], b[:, :lo]) and torch.equal(out[:, hi:], b[:, hi:])
    assert a == pytest.approx(a_full, rel=1e-6)
    np.testing.assert_allclose(y.cpu().numpy(), y_full.cpu().numpy(), rtol=0, atol=1e-4)
    for e, *_ in engs:
        e.close()
    full.close()
