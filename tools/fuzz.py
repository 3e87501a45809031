"""dev helper (GPU box): the seeded shape sweep of tests/test_gpu_parity.py::test_random_shapes_against_oracle with
many more cases (python tools/fuzz.py [cases] [seed])"""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
from synth import synth_frame, synth_watermark

wm = importlib.import_module("watermarking-gpu_amd")


def run(n, seed=7, verbose=True):
    """n seeded random cases against the oracle; returns the number of failing cases"""
    rng = np.random.default_rng(seed)
    bad = 0
    for case in range(n):
        bad += one_case(rng, case, verbose)
    return bad


def one_case(rng, case, verbose):
    bad = 0
    if True:
        R = int(rng.integers(64, 500))
        Cc = int(rng.choice([rng.integers(64, 1100), 4 * rng.integers(16, 280), 256 * rng.integers(1, 5) + rng.integers(0, 8)]))
        u8 = bool(rng.integers(0, 2))
        nvf = not rng.integers(0, 3)
        p = int(rng.choice([3, 5, 7, 9])) if nvf else 3
        mk, omk = (wm.MASK_TYPE.NVF, O.MASK_NVF) if nvf else (wm.MASK_TYPE.ME, O.MASK_ME)
        F = int(rng.integers(1, 10))   # frames per launch: 4 and more take the frame-quad mapping
        xs = np.stack([synth_frame(R, Cc, frame=case * 16 + f, dtype=np.uint8 if u8 else np.float32) for f in range(F)])
        W = synth_watermark(R, Cc)
        eng = wm.Watermark(R, Cc, W, p, 40.0, nslots=1, max_frames=F)
        rps = int(rng.integers(5, 90)) if rng.integers(0, 2) else 0
        if rps:
            eng.set_rows_per_segment(rps)
        xd = torch.from_numpy(xs).cuda()
        ys, as_ = eng.makeWatermark(xd, xd, mk)
        fchk = int(rng.integers(0, F))   # one frame of the batch against the oracle
        x, y, a = xs[fchk], ys[fchk], as_[fchk]
        tag = f"case {case}: {R}x{Cc} {'u8' if u8 else 'f32'} mask={int(mk)} p={p} rps={rps} F={F} frame {fchk}"
        try:
            if u8:
                so, yo, ao = O.embed_u8(x, W, p=p, mask=omk)
                d = np.abs(y.cpu().numpy().astype(int) - yo.astype(int))
                assert d.max() <= 1 and (d != 0).mean() <= 2e-3, "y"
                cref = O.detect_u8(yo, W, p=p, mask=omk)[1]
            else:
                so, yo, ao = O.embed(x, x, W, p=p, mask=omk)
                assert np.abs(y.cpu().numpy() - yo).max() <= 1e-3, "y"
                cref = O.detect(yo, W, p=p, mask=omk)[1]
            assert abs(a - ao) <= 1e-4 * abs(ao), "a"
            yb = ys.clone()
            yb[fchk] = torch.from_numpy(yo).cuda()
            c = eng.detectWatermark(yb, mk)[fchk]
            assert abs(c - cref) <= 1e-5, f"corr {c} {cref}"
            if not u8 and p == 3:
                # the Gram hand-over (wm_set_handover): embed on a slot, then the Gram sums and the score through WM_MEM_SLOT_OUT
                # against the same plane handed in as an ordinary device plane (whatever the shape: shapes the hand-over does
                # not cover must take the ordinary sweep silently)
                import ctypes as C
                # (the reference for "the hand-over changes nothing" is the same embed on the sweeps WITHOUT it: `ys` above may
                # come from the fused one-frame kernel, whose strength can differ from the sweeps' in the last bits)
                y1 = torch.empty_like(xd)
                eng.embed_async(xd, xd, y1, mk, 0)
                eng.sync(0)
                eng.set_handover(True)
                y2 = torch.empty_like(xd)
                eng.embed_async(xd, xd, y2, mk, 0)
                sp = wm.wm_plane(None, R, Cc, 1, wm.WM_F32, wm.WM_MEM_SLOT_OUT, F, Cc, 0, R * Cc)
                buf = (C.c_double * (44 * F))()
                assert wm.lib().wm_gram(eng._ctx, C.byref(sp), buf, 0) == 0, "wm_gram(SLOT_OUT)"
                t_ho = np.array(buf[:], dtype=np.float64).reshape(F, 44)
                c_ho, c_in = (C.c_float * F)(), (C.c_float * F)()
                eng.detect_async(sp, mk, 0, corr_out=c_ho)
                eng.sync(0)
                assert torch.equal(y2, y1), "hand-over changed y"
                assert float((y1.float() - ys.float()).abs().max()) <= 1e-3, "fused / sweeps y"
                t_in = eng.gram_totals(y2).reshape(F, 44)
                eng.detect_async(y2, mk, 0, corr_out=c_in)
                eng.sync(0)
                sc = np.abs(t_in).max(axis=1, keepdims=True)
                assert np.abs(t_ho / sc - t_in / sc).max() <= 2e-15, f"hand-over Gram sums {np.abs(t_ho / sc - t_in / sc).max()}"
                assert max(abs(u - v) for u, v in zip(c_ho, c_in)) <= 2e-7, "hand-over score"
        except AssertionError as e:
            bad += 1
            if verbose:
                print("FAIL", tag, e, flush=True)
        eng.close()
    return bad


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    bad = run(n, int(sys.argv[2]) if len(sys.argv) > 2 else 7)
    print(f"{n} cases, {bad} failures")
    sys.exit(1 if bad else 0)
