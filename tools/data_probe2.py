"""dev helper (GPU box): is the slow case of data_probe.py a property of the data or of the moment?  One engine (W = torch.randn),
the bench's synthetic frames and a low-activity set, alternating, several rounds, with pauses"""
import ctypes as C
import importlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
wm = importlib.import_module("watermarking-gpu_amd")
synth = importlib.import_module("watermarking-gpu_amd.synth")
R, Cc, F = 2160, 3840, 16
dev = torch.device("cuda", 0)
a, corr = (C.c_float * F)(), (C.c_float * F)()
Wr = torch.randn((R, Cc), generator=torch.Generator().manual_seed(2))
order = sys.argv[1] if len(sys.argv) > 1 else "rgh"
mk = {"r": ("randn", lambda: wm.Watermark(R, Cc, Wr.numpy(), 3, 40.0, nslots=1, max_frames=F)),
      "g": ("generated", lambda: wm.Watermark.generated(R, Cc, synth.SEED, 3, 40.0, nslots=1, max_frames=F)),
      "h": ("randn*0.5", lambda: wm.Watermark(R, Cc, (0.5 * Wr).numpy(), 3, 40.0, nslots=1, max_frames=F))}
ndummy = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dummies = [torch.cuda.Stream() for _ in range(ndummy)]     # HIP streams created BEFORE the first engine's
for d in dummies:
    with torch.cuda.stream(d):
        torch.zeros(16, device=dev).add_(1)
torch.cuda.synchronize()
engs = {}
for ch in order:
    engs[mk[ch][0] + f"#{len(engs)}"] = mk[ch][1]()
xs = {"synth": synth.synth_frames_torch(R, Cc, F, dev), "const+noise": (128 + 20 * torch.randn((F, R, Cc), device=dev)).clamp(0, 255)}
y = torch.empty_like(xs["synth"])
torch.cuda.synchronize()
for rnd in range(2):
    for wn, eng in engs.items():
        for xn, x in xs.items():
            px, py = wm.plane_of(x), wm.plane_of(y)
            eng.prof_enable(True); eng.prof_reset()
            for _ in range(10):
                eng.embed_async(px, px, py, 0, 0, a_out=a); eng.detect_async(py, 0, 0, corr_out=corr)
            eng.sync(0)
            rep = eng.prof_report(); eng.prof_enable(False)
            print(f"round {rnd} W {wn:12s} x {xn:12s} a={a[0]:7.3f} | " + "  ".join(f"{kn}:{1e3 * ms / cnt:.1f}" for kn, (cnt, ms) in rep.items()), flush=True)
    time.sleep(0.5)
