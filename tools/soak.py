"""dev helper (GPU box): soak of the batched pipeline.  Every slot alternates between two input batches, so the partial
records a sweep leaves in the slot's scratch differ from the ones the next op writes; every strength, correlation and
output plane must be bitwise equal to the first result of the same batch.  Catches a fold tail that reads a stale
partial (the partials of one frame are written on several XCDs), a ticket left non-zero, or a race between slots.
usage: python tools/soak.py [steps]"""
import ctypes as C
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
wm = importlib.import_module("watermarking-gpu_amd")
from quick_bench import fake_frames


def soak(rows, cols, F, S, steps, dtype, mask):
    W = torch.randn((rows, cols), generator=torch.Generator().manual_seed(2)).numpy()
    eng = wm.Watermark(rows, cols, W, 3, 40.0, nslots=S, max_frames=F)
    base = [fake_frames(rows, cols, F, dtype) for _ in range(S)]
    # batch B of every slot: the same frames in reverse order and flipped upside down (other sums in every block)
    xsets = [base, [torch.flip(x, dims=(0, 1)).contiguous() for x in base]]
    ys = [torch.empty_like(x) for x in base]
    a = [(C.c_float * F)() for _ in range(S)]
    corr = [(C.c_float * F)() for _ in range(S)]
    pxs = [[wm.plane_of(x) for x in xs] for xs in xsets]
    py = [wm.plane_of(y) for y in ys]
    torch.cuda.synchronize()
    refs = [None, None]
    bad = 0
    for it in range(steps):
        which = it & 1
        px = pxs[which]
        ref = refs[which]
        for s in range(S):
            eng.embed_async(px[s], px[s], py[s], mask, s, a_out=a[s])
            eng.detect_async(py[s], mask, s, corr_out=corr[s])
        for s in range(S):
            eng.sync(s)
        cur = ([list(v) for v in a], [list(v) for v in corr])
        if it % 50 < 2 or ref is None:
            sums = [int(y.view(torch.uint8).to(torch.int64).sum()) for y in ys]  # byte checksum of the outputs
            cur = cur + (sums,)
        if ref is None:
            refs[which] = cur
        else:
            if cur[0] != ref[0] or cur[1] != ref[1] or (len(cur) > 2 and cur[2] != ref[2]):
                bad += 1
                print("MISMATCH at step", it, flush=True)
    eng.close()
    print(f"{rows}x{cols} {str(dtype)[6:]} mask={mask} F={F} S={S}: {steps} steps, {bad} mismatching steps", flush=True)
    return bad


if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    bad = soak(2160, 3840, 16, 3, steps, torch.float32, 0)
    bad += soak(2160, 3840, 16, 3, steps, torch.uint8, 0)
    bad += soak(1080, 1920, 8, 4, steps, torch.float32, 1)
    bad += soak(300, 700, 5, 4, 4 * steps, torch.float32, 0)
    sys.exit(1 if bad else 0)
