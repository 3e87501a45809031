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


@pytest.mark.parametrize("mask", ["ME", "NVF"])
def test_band_device_resident_exchange_matches_the_host_exchange(wm, tc, mask):
    """wm_band_*_dev: the same phases with the totals left in device memory.  Three bands in one process, the collectives
    emulated by tensor operations on the device (sum / concatenate): every band must end with the SAME strength and
    correlation as the host-exchange calls give, and the stitched output must be identical"""
    torch = tc
    bands = importlib.import_module("watermarking-gpu_amd.bands")
    R, Cc, world = 190, 772, 3
    x = synth_frame(R, Cc, frame=6)
    W = synth_watermark(R, Cc)
    mk = wm.MASK_TYPE[mask]
    xd = torch.from_numpy(x).cuda()
    engs = []
    for r in range(world):
        g0, g1, lo, hi = bands.band_with_halo(R, r, world)
        e = wm.Watermark(g1 - g0, Cc, np.ascontiguousarray(W[g0:g1]), 3, 40.0)
        e.band_configure(lo, hi, R)
        e.set_stream_current()
        engs.append((e, g0, g1, lo, hi))
    f64 = dict(dtype=torch.float64, device="cuda")
    # host exchange (reference)
    tot_h = sum(e.gram_totals(xd[g0:g1].contiguous()) for (e, g0, g1, lo, hi) in engs)
    for (e, *_) in engs:
        assert e.band_solve(tot_h) == 0
    st = [e.band_stats(xd[g0:g1].contiguous(), mk) for (e, g0, g1, lo, hi) in engs]
    mx, ss = max(s[0] for s in st), sum(s[1] for s in st)
    y_h = torch.empty_like(xd)
    for (e, g0, g1, lo, hi) in engs:
        v = xd[g0:g1].contiguous(); out = v.clone()
        a_h = e.band_embed(v, v, out, mk, mx, ss)
        y_h[g0 + lo:g0 + hi] = out[lo:hi]
    # device-resident exchange: no host copy between the phases
    tots = [torch.zeros(44, **f64) for _ in engs]
    for (e, g0, g1, lo, hi), t in zip(engs, tots):
        e.band_gram_dev(xd[g0:g1].contiguous(), t)
    tot_d = torch.stack(tots).sum(0)                       # what the all-reduce leaves on every rank
    np.testing.assert_allclose(tot_d.cpu().numpy(), tot_h, rtol=1e-14)
    parts = torch.zeros(2 * world, **f64)
    for k, (e, g0, g1, lo, hi) in enumerate(engs):
        e.band_solve_dev(tot_d)
        ms = torch.zeros(2, **f64)
        e.band_stats_dev(xd[g0:g1].contiguous(), mk, ms)
        parts[2 * k:2 * k + 2] = ms                       # what the all-gather leaves on every rank
    y_d = torch.empty_like(xd)
    a_d = []
    for (e, g0, g1, lo, hi) in engs:
        v = xd[g0:g1].contiguous(); out = v.clone()
        a_dev = torch.zeros(1, dtype=torch.float32, device="cuda")
        e.band_embed_dev(v, v, out, mk, parts, world, a_dev)
        y_d[g0 + lo:g0 + hi] = out[lo:hi]
        a_d.append(float(a_dev.item()))
    assert a_d[0] == a_d[1] == a_d[2] == a_h
    assert torch.equal(y_d, y_h)
    # detect on the stitched image
    toty = [torch.zeros(44, **f64) for _ in engs]
    for (e, g0, g1, lo, hi), t in zip(engs, toty):
        e.band_gram_dev(y_d[g0:g1].contiguous(), t)
    toty_d = torch.stack(toty).sum(0)
    sums = []
    for (e, g0, g1, lo, hi) in engs:
        e.band_solve_dev(toty_d)
        sm = torch.zeros(3, **f64)
        e.band_detect_sums_dev(y_d[g0:g1].contiguous(), mk, sm)
        sums.append(sm)
    sums_d = torch.stack(sums).sum(0)
    cs = []
    for (e, *_) in engs:
        c = torch.zeros(1, dtype=torch.float32, device="cuda")
        e.band_corr_dev(sums_d, c)
        cs.append(float(c.item()))
    full = wm.Watermark(R, Cc, W, 3, 40.0)
    assert cs[0] == cs[1] == cs[2] == pytest.approx(full.detectWatermark(y_d, mk), abs=2e-6)
    assert cs[0] == pytest.approx(O.detect(y_d.cpu().numpy(), W, mask=O.MASK_ME if mask == "ME" else O.MASK_NVF)[1], abs=1e-5)
    # an unsolvable band set: the strength comes back as NaN on the device, the output is the input, the correlation 0
    flat = torch.full((engs[0][2] - engs[0][1], Cc), 77.0, device="cuda")
    e0 = engs[0][0]
    t0 = torch.zeros(44, **f64)
    e0.band_gram_dev(flat, t0); e0.band_solve_dev(t0)
    ms = torch.zeros(2, **f64); e0.band_stats_dev(flat, wm.MASK_TYPE.ME, ms)
    out = torch.zeros_like(flat); a_dev = torch.zeros(1, dtype=torch.float32, device="cuda")
    e0.band_embed_dev(flat, flat, out, wm.MASK_TYPE.ME, ms, 1, a_dev)
    lo, hi = engs[0][3], engs[0][4]
    assert np.isnan(float(a_dev.item())) and torch.equal(out[lo:hi], flat[lo:hi])
    for e, *_ in engs:
        e.close()
    full.close()


CHILD_RCCL_BANDS = r"""
import importlib, json, os, sys
import numpy as np
sys.path.insert(0, os.environ["WM_ROOT"]); sys.path.insert(0, os.path.join(os.environ["WM_ROOT"], "tests"))
import torch
import torch.distributed as dist
wm = importlib.import_module("watermarking-gpu_amd")
bands = importlib.import_module("watermarking-gpu_amd.bands")
from synth import synth_frame, synth_watermark
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
R, C = 203, 508
x = synth_frame(R, C, frame=3); W = synth_watermark(R, C)
bw = bands.BandedWatermark(R, C, W, 3, 40.0, 0, 1, device=0)
assert bw.on_device
bw.force_collective = True                               # RCCL all-reduce / all-gather with one rank
band = torch.from_numpy(x).cuda()
full = wm.Watermark(R, C, W, 3, 40.0)
res = {}
for name in ("ME", "NVF"):
    mk = wm.MASK_TYPE[name]
    yb, a = bw.embed(band, mk)
    yb = bw.exchange_halos(yb)
    c = bw.detect(yb, mk)
    yf, af = full.makeWatermark(band, band, mk)
    res[name] = {"a": a, "a_full": af, "c": c, "c_full": full.detectWatermark(yf, mk), "max_dy": float((yb - yf).abs().max())}
maps = open("/proc/self/maps").read()
res["rccl_loaded"] = any("librccl" in l for l in maps.splitlines())
print("RESULT " + json.dumps(res))
bw.close(); full.close()
dist.destroy_process_group()
"""


def test_banded_engine_device_resident_over_rccl_world1():
    """bands.BandedWatermark with backend "nccl": the exchange stays on the device (wm_band_*_dev on torch's current stream,
    RCCL all-reduce of 44 doubles / all-gather of {max, sum} / all-reduce of 3 sums, forced with the one rank this box has)
    and gives the whole-image engine's results"""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WM_ROOT=root, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29400 + os.getpid() % 500), HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", CHILD_RCCL_BANDS], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-4000:]
    res = json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    assert res["rccl_loaded"]
    for name in ("ME", "NVF"):
        r = res[name]
        assert r["a"] == pytest.approx(r["a_full"], rel=1e-6) and r["c"] == pytest.approx(r["c_full"], abs=2e-6) and r["max_dy"] <= 1e-4, r
