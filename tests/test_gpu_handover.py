"""Gram hand-over from an embed to the detector that reads its output (wm.h wm_set_handover): k_embed leaves the lag sums of y
that stay inside each wavefront's tile, k_gram_ho adds the products across tile seams, the border frame and the solve, and the
detector's own Gram sweep over y is not run.  Checked here: the 44 sums against k_gram's over the same plane (the same exact
products in another f64 summation order: 1e-14 relative; equal bits are counted, not required), the detector's score against
the independent path and the oracle, over the tile geometries the launch can produce (segments of 2..48 rows, shifted last
strip, one strip, one segment, batches with and without the 4-frames-per-block mapping), both masks, a separate base plane,
in-place frames, an unsolvable frame (the passthrough plane is what the detector reads) -- and that everything the hand-over
does not cover (u8 planes, RGB bases, widths off the aligned path, another plane than the slot's output) silently takes the
ordinary sweep."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
from synth import synth_frame, synth_watermark

pytestmark = pytest.mark.gpu

TOL_CORR = 1e-5


@pytest.fixture(scope="module")
def tc():
    import torch
    assert torch.cuda.is_available()
    return torch


def slot_plane(wm, R, Cc, frames, dtype=None):
    return wm.wm_plane(None, R, Cc, 1, wm.WM_F32 if dtype is None else dtype, wm.WM_MEM_SLOT_OUT, frames, Cc, 0, R * Cc)


def totals_of_slot(wm, eng, R, Cc, frames):
    buf = (C.c_double * (44 * frames))()
    sp = slot_plane(wm, R, Cc, frames)
    assert wm.lib().wm_gram(eng._ctx, C.byref(sp), buf, 0) == 0
    return np.array(buf[:], dtype=np.float64).reshape(frames, 44)


def run_pair(wm, torch, R, Cc, F, mask, rps=0, base_is_x=True, inplace=False, frames_np=None):
    """embed (slot 0, hand-over on) then detect on WM_MEM_SLOT_OUT; returns everything the checks need"""
    mk = wm.MASK_TYPE.ME if mask == "ME" else wm.MASK_TYPE.NVF
    W = synth_watermark(R, Cc)
    eng = wm.Watermark(R, Cc, W, 3, 40.0, nslots=2, max_frames=F)
    ref = wm.Watermark(R, Cc, W, 3, 40.0, nslots=2, max_frames=F)
    eng.set_handover(True)
    if rps:
        eng.set_rows_per_segment(rps)
        ref.set_rows_per_segment(rps)
    xs = np.stack([synth_frame(R, Cc, frame=f) for f in range(F)]) if frames_np is None else frames_np
    x = torch.from_numpy(xs).cuda()
    base = x if base_is_x else torch.from_numpy(np.stack([synth_frame(R, Cc, frame=100 + f) for f in range(F)])).cuda()
    y = x if inplace else torch.empty_like(x)
    x_keep = x.clone()
    a, st = (C.c_float * F)(), (C.c_int * F)()
    corr, corr_ref = (C.c_float * F)(), (C.c_float * F)()
    eng.prof_enable(True)
    eng.embed_async(x, base, y, mk, 0, a_out=a, status_out=st)
    tot_ho = totals_of_slot(wm, eng, R, Cc, F)          # (synchronises the slot)
    sp = slot_plane(wm, R, Cc, F)
    eng.detect_async(sp, mk, 0, corr_out=corr)
    eng.sync(0)
    rep = eng.prof_report()
    # the independent path on the same plane
    tot_ref = ref.gram_totals(y).reshape(F, 44)
    ref.detect_async(y, mk, 0, corr_out=corr_ref)
    ref.sync(0)
    # ... and the same embed without the hand-over writes the same plane
    y2 = torch.empty_like(x_keep)
    a2 = (C.c_float * F)()
    ref.embed_async(x_keep, x_keep if base_is_x else base, y2, mk, 0, a_out=a2)
    ref.sync(0)
    out = dict(eng=eng, ref=ref, W=W, xs=xs, y=y, y2=y2, a=list(a), a2=list(a2), st=list(st), corr=list(corr), corr_ref=list(corr_ref),
               tot_ho=tot_ho, tot_ref=tot_ref, rep=rep)
    return out


def check(o, expect_ho=True):
    assert np.array_equal(o["y"].cpu().numpy(), o["y2"].cpu().numpy()), "the hand-over changed the embedded plane"
    assert o["a"] == o["a2"]
    if expect_ho:
        assert "k_gram_ho" in o["rep"], o["rep"]
        assert o["rep"]["k_gram_ho"][0] == 2      # wm_gram + wm_detect on the slot's output
    else:
        assert "k_gram_ho" not in o["rep"], o["rep"]
    scale = np.abs(o["tot_ref"]).max(axis=1, keepdims=True)
    np.testing.assert_allclose(o["tot_ho"] / scale, o["tot_ref"] / scale, rtol=0, atol=2e-15)
    for c1, c2 in zip(o["corr"], o["corr_ref"]):
        assert c1 == pytest.approx(c2, abs=2e-7)
    o["eng"].close(); o["ref"].close()


# (rows, cols, frames, rows per segment): one strip; 2 strips + shifted last strip (516, 1920); one segment; segments of 2, 3, 5
# rows (every row is next to a seam); a last segment of one row (97 % 8 == 1); batches below and above the 4-frame mapping;
# the reference's 4k_non_divisible width (3872 = 15 strips + 32 columns) and BASELINE.json's 8K
CASES = [(64, 256, 2, 0), (100, 512, 3, 0), (97, 516, 4, 8), (97, 516, 5, 2), (130, 1028, 2, 3), (57, 772, 6, 5), (40, 260, 4, 40),
         (270, 1024, 8, 0), (1080, 1920, 4, 0), (2160, 3840, 2, 0), (2160, 3872, 2, 0), (4320, 7680, 2, 0)]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("mask", ["ME", "NVF"])
def test_handover_totals_and_score(wm, tc, case, mask):
    R, Cc, F, rps = case
    o = run_pair(wm, tc, R, Cc, F, mask, rps=rps)
    same = int((o["tot_ho"] == o["tot_ref"]).sum())
    print(f"{R}x{Cc} F={F} rps={rps} {mask}: {same}/{o['tot_ho'].size} Gram sums bit-equal to k_gram's")
    if R * Cc <= 270 * 1024:
        omk = O.MASK_ME if mask == "ME" else O.MASK_NVF
        for f in range(F):
            yo = o["y"][f].cpu().numpy()
            assert o["corr"][f] == pytest.approx(O.detect(yo, o["W"], mask=omk)[1], abs=TOL_CORR)
    check(o)


def test_handover_separate_base_and_in_place(wm, tc):
    o = run_pair(wm, tc, 130, 516, 4, "ME", base_is_x=False)
    check(o)
    # in place (the video contract, main.cpp:356,380): input, base and output are one plane
    torch = tc
    R, Cc, F = 130, 516, 4
    W = synth_watermark(R, Cc)
    eng = wm.Watermark(R, Cc, W, 3, 40.0, nslots=2, max_frames=F)
    ref = wm.Watermark(R, Cc, W, 3, 40.0, nslots=2, max_frames=F)
    eng.set_handover(True)
    xs = np.stack([synth_frame(R, Cc, frame=f) for f in range(F)])
    fr, fr2 = torch.from_numpy(xs).cuda(), torch.from_numpy(xs).cuda()
    corr, corr2 = (C.c_float * F)(), (C.c_float * F)()
    eng.prof_enable(True)
    eng.embed_async(fr, fr, fr, wm.MASK_TYPE.ME, 0)
    eng.detect_async(slot_plane(wm, R, Cc, F), wm.MASK_TYPE.ME, 0, corr_out=corr)
    eng.sync(0)
    ref.embed_async(fr2, fr2, fr2, wm.MASK_TYPE.ME, 0)
    ref.detect_async(fr2, wm.MASK_TYPE.ME, 0, corr_out=corr2)
    ref.sync(0)
    assert torch.equal(fr, fr2)
    assert "k_gram_ho" in eng.prof_report()
    for c1, c2 in zip(corr, corr2):
        assert c1 == pytest.approx(c2, abs=2e-7)
    eng.close(); ref.close()


def test_handover_unsolvable_frame_hands_over_the_passthrough_plane(wm, tc):
    """a constant frame has a singular system: the embed passes the base through (Watermark.cpp:164-165) and that plane is what
    the detector reads -- its lag sums must be the base's, whatever the coefficients of the failed solve hold"""
    R, Cc, F = 64, 512, 4
    xs = np.stack([synth_frame(R, Cc, frame=f) for f in range(F)])
    xs[2] = 77.0
    o = run_pair(wm, tc, R, Cc, F, "ME", base_is_x=False, frames_np=xs)
    assert o["st"][2] != 0 and o["st"][0] == 0
    check(o)


def test_handover_is_deterministic_and_independent_of_what_ran_before(wm, tc):
    """every fold of the hand-over runs in a fixed order (per lane, per wave, wave records in index order through the seam
    blocks, seam records in index order): the same frames give the same bits, run to run, on either slot, and whatever other
    batch used the slot's record arrays in between"""
    torch = tc
    R, Cc, F = 270, 1028, 8
    W = synth_watermark(R, Cc)
    eng = wm.Watermark(R, Cc, W, 3, 40.0, nslots=2, max_frames=F)
    eng.set_handover(True)
    xa = torch.from_numpy(np.stack([synth_frame(R, Cc, frame=f) for f in range(F)])).cuda()
    xb = torch.from_numpy(np.stack([synth_frame(R, Cc, frame=50 + f) for f in range(F)])).cuda()
    y = torch.empty_like(xa)
    sp = slot_plane(wm, R, Cc, F)
    first = None
    for rnd in range(6):
        sl = rnd % 2
        corr = (C.c_float * F)()
        eng.embed_async(xa, xa, y, wm.MASK_TYPE.ME, sl)
        buf = (C.c_double * (44 * F))()
        eng.detect_async(sp, wm.MASK_TYPE.ME, sl, corr_out=corr)
        eng.sync(sl)
        got = (list(corr), y.clone())
        if first is None:
            first = got
        assert got[0] == first[0] and torch.equal(got[1], first[1]), f"round {rnd} differs"
        # another batch through the same slot's arrays
        eng.embed_async(xb, xb, y, wm.MASK_TYPE.NVF, sl)
        eng.detect_async(sp, wm.MASK_TYPE.NVF, sl, corr_out=corr)
        eng.sync(sl)
    eng.close()


def test_whatever_the_handover_does_not_cover_takes_the_gram_sweep(wm, tc):
    torch = tc
    L = wm.lib()
    # (a) a width off the aligned path, (b) fewer than 256 columns: no hand-over instantiation, wm_detect(SLOT_OUT) still right
    for R, Cc in ((64, 518), (80, 200)):
        o = run_pair(wm, tc, R, Cc, 4, "ME")
        check(o, expect_ho=False)
    # a single frame: a one-frame k_gram_ho is as latency-bound as the k_gram it would replace -- not handed over
    o = run_pair(wm, tc, 64, 512, 1, "ME")
    check(o, expect_ho=False)
    # (c) u8 planes
    R, Cc, F = 64, 512, 4
    W = synth_watermark(R, Cc)
    eng = wm.Watermark(R, Cc, W, 3, 40.0, nslots=2, max_frames=F)
    eng.set_handover(True)
    eng.prof_enable(True)
    xs = torch.from_numpy(np.stack([synth_frame(R, Cc, frame=f, dtype=np.uint8) for f in range(F)])).cuda()
    ys = torch.empty_like(xs)
    corr, corr2 = (C.c_float * F)(), (C.c_float * F)()
    eng.embed_async(xs, xs, ys, wm.MASK_TYPE.ME, 0)
    eng.detect_async(slot_plane(wm, R, Cc, F, wm.WM_U8), wm.MASK_TYPE.ME, 0, corr_out=corr)
    eng.sync(0)
    eng.detect_async(ys, wm.MASK_TYPE.ME, 0, corr_out=corr2)
    eng.sync(0)
    assert list(corr) == list(corr2) and "k_gram_ho" not in eng.prof_report()
    # (d) f32 again on the same context: the hand-over applies; then ANOTHER plane than the slot's output: ordinary sweep; then
    # an embed with an RGB base replaces the slot's output: its detector input is not a grey plane of the slot any more
    x = torch.from_numpy(np.stack([synth_frame(R, Cc, frame=f) for f in range(F)])).cuda()
    y = torch.empty_like(x)
    eng.prof_reset()
    eng.embed_async(x, x, y, wm.MASK_TYPE.ME, 0)
    eng.detect_async(slot_plane(wm, R, Cc, F), wm.MASK_TYPE.ME, 0, corr_out=corr)
    eng.sync(0)
    assert eng.prof_report()["k_gram_ho"][0] == 1
    eng.prof_reset()
    eng.detect_async(y, wm.MASK_TYPE.ME, 0, corr_out=corr2)
    eng.sync(0)
    assert "k_gram_ho" not in eng.prof_report()
    for c1, c2 in zip(corr, corr2):
        assert c1 == pytest.approx(c2, abs=2e-7)
    rgb = torch.stack([x, x, x], dim=1).contiguous()
    out = torch.empty_like(rgb)
    eng.embed_async(x, rgb, out, wm.MASK_TYPE.ME, 0)
    sp = slot_plane(wm, R, Cc, F)
    assert L.wm_detect(eng._ctx, 0, C.byref(sp), corr, None, 0) == wm.WM_ERR_BAD_ARG
    eng.sync(0)
    # (e) switching the hand-over off forgets what a slot holds
    eng.embed_async(x, x, y, wm.MASK_TYPE.ME, 0)
    eng.sync(0)
    eng.set_handover(False)
    eng.prof_reset()
    eng.detect_async(slot_plane(wm, R, Cc, F), wm.MASK_TYPE.ME, 0, corr_out=corr)
    eng.sync(0)
    assert "k_gram_ho" not in eng.prof_report() and "k_gram" in eng.prof_report()
    eng.close()


def test_handover_verify_mode_catches_a_plane_modified_behind_the_embed(wm, tc, monkeypatch):
    """WM_HANDOVER_VERIFY=1 (wm.h, the hazard note of wm_set_handover): the handed-over Gram totals are held against an
    ordinary Gram sweep over the plane as it is.  Untouched plane: the call passes and scores as without the mode.  One pixel
    of the output changed from ANOTHER stream between embed and detect -- the write the library cannot see -- : the
    detector fails with WM_ERR_RUNTIME and names the frame, instead of correlating old sums with new pixels."""
    torch = tc
    monkeypatch.setenv("WM_HANDOVER_VERIFY", "1")
    R, Cc, F = 130, 516, 4
    W = synth_watermark(R, Cc)
    eng = wm.Watermark(R, Cc, W, 3, 40.0, nslots=2, max_frames=F)
    monkeypatch.delenv("WM_HANDOVER_VERIFY")
    plain = wm.Watermark(R, Cc, W, 3, 40.0, nslots=2, max_frames=F)
    for e in (eng, plain):
        e.set_handover(True)
    x = torch.from_numpy(np.stack([synth_frame(R, Cc, frame=f) for f in range(F)])).cuda()
    sp = slot_plane(wm, R, Cc, F)
    L = wm.lib()
    scores = {}
    for name, e in (("verify", eng), ("plain", plain)):
        y = torch.empty_like(x)
        a, corr = (C.c_float * F)(), (C.c_float * F)()
        e.prof_enable(True)
        e.embed_async(x, x, y, wm.MASK_TYPE.ME, 0, a_out=a)
        e.detect_async(sp, wm.MASK_TYPE.ME, 0, corr_out=corr)
        e.sync(0)
        assert "k_gram_ho" in e.prof_report(), "the hand-over kernels did not run"
        scores[name] = list(corr)
    assert scores["verify"] == scores["plain"]
    # now the hazard: a write to the output plane from torch's stream, after the embed has completed
    y = torch.empty_like(x)
    a, corr = (C.c_float * F)(), (C.c_float * F)()
    eng.embed_async(x, x, y, wm.MASK_TYPE.ME, 0, a_out=a)
    eng.sync(0)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        y[2, 77, 300] += 3.0
    side.synchronize()
    rc = L.wm_detect(eng._ctx, int(wm.MASK_TYPE.ME), C.byref(sp), corr, None, 0)
    assert rc == wm.WM_ERR_RUNTIME, rc
    msg = L.wm_last_error(eng._ctx).decode()
    assert "WM_HANDOVER_VERIFY" in msg and "frame 2" in msg, msg
    # the same write without the mode goes unnoticed (that is the hazard): the call succeeds with a score built on stale sums
    y2 = torch.empty_like(x)
    plain.embed_async(x, x, y2, wm.MASK_TYPE.ME, 0, a_out=a)
    plain.sync(0)
    with torch.cuda.stream(side):
        y2[2, 77, 300] += 3.0
    side.synchronize()
    assert L.wm_detect(plain._ctx, int(wm.MASK_TYPE.ME), C.byref(sp), corr, None, 0) == 0
    plain.sync(0)
    # after the failure the slot's hand-over is gone: the next detect on the slot's output takes the ordinary sweep and is right
    corr2 = (C.c_float * F)()
    assert L.wm_detect(eng._ctx, int(wm.MASK_TYPE.ME), C.byref(sp), corr2, None, 0) == 0
    eng.sync(0)
    ref = (C.c_float * F)()
    plain.set_handover(False)
    plain.detect_async(y, wm.MASK_TYPE.ME, 1, corr_out=ref)
    plain.sync(1)
    assert max(abs(p - q) for p, q in zip(corr2, ref)) <= 1e-6
    eng.close(); plain.close()
