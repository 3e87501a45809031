"""dev helper (GPU box): per-kernel event times of the serialised 16-frame 4K pipeline for different INPUT DATA (the bench's
synthetic frames against quick_bench's) and different W (generated on the device against torch.randn)"""
import ctypes as C
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
wm = importlib.import_module("watermarking-gpu_amd")
synth = importlib.import_module("watermarking-gpu_amd.synth")
from quick_bench import fake_frames  # noqa: E402

import glob


def clocks():
    out = []
    for name in ("freq1_input", "freq2_input", "power1_average"):
        v = []
        for pth in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/" + name):
            try:
                v.append(float(open(pth).read()) / 1e6)
            except Exception:
                pass
        out.append(max(v) if v else float("nan"))
    for name in ("pp_dpm_mclk", "pp_dpm_fclk", "pp_dpm_socclk"):
        cur = []
        for pth in glob.glob("/sys/class/drm/card*/device/" + name):
            try:
                cur += [l.split(":")[1].strip().rstrip("*").strip() for l in open(pth).read().splitlines() if l.strip().endswith("*")]
            except Exception:
                pass
        out.append("/".join(sorted(set(cur))) if cur else "-")
    return f"sclk {out[0]:.0f} mclk(freq2) {out[1]:.0f} power {out[2]:.0f} W  mclk {out[3]} fclk {out[4]} socclk {out[5]}"


R, Cc, F = 2160, 3840, 16
dev = torch.device("cuda", 0)
a, corr = (C.c_float * F)(), (C.c_float * F)()
# frame buffers allocated ONCE, first thing (fresh, large allocations), and refilled in place: separates the effect of WHERE
# the planes live from the effect of WHAT they hold
fixed = len(sys.argv) > 1 and sys.argv[1] == "fixed"
if fixed:
    xbuf = torch.empty((F, R, Cc), dtype=torch.float32, device=dev)
    ybuf = torch.empty_like(xbuf)
for wname in ("generated", "randn"):
    if wname == "generated":
        eng = wm.Watermark.generated(R, Cc, synth.SEED, 3, 40.0, nslots=1, max_frames=F)
    else:
        eng = wm.Watermark(R, Cc, torch.randn((R, Cc), generator=torch.Generator().manual_seed(2)).numpy(), 3, 40.0, nslots=1, max_frames=F)
    for xname in ("synth", "fake", "synth_rounded", "constant+noise"):
        if xname == "synth":
            x = synth.synth_frames_torch(R, Cc, F, dev)
        elif xname == "fake":
            x = fake_frames(R, Cc, F)
        elif xname == "synth_rounded":
            x = synth.synth_frames_torch(R, Cc, F, dev).round()
        else:
            x = (128 + 20 * torch.randn((F, R, Cc), device=dev)).clamp(0, 255)
        if fixed:
            xbuf.copy_(x); del x
            x, y = xbuf, ybuf
        else:
            y = torch.empty_like(x)
        px, py = wm.plane_of(x), wm.plane_of(y)
        for _ in range(3):
            eng.embed_async(px, px, py, 0, 0, a_out=a); eng.detect_async(py, 0, 0, corr_out=corr)
        eng.sync(0)
        eng.prof_enable(True); eng.prof_reset()
        for _ in range(10):
            eng.embed_async(px, px, py, 0, 0, a_out=a); eng.detect_async(py, 0, 0, corr_out=corr)
        eng.sync(0)
        ck = clocks()
        rep = eng.prof_report(); eng.prof_enable(False)
        print(f"W {wname:9s} x {xname:15s} a={a[0]:8.3f} corr={corr[0]:.4f} | " + "  ".join(f"{kn}:{1e3 * ms / cnt:.1f}us" for kn, (cnt, ms) in rep.items()), flush=True)
        if not fixed:
            del x, y
        print(f"      {ck}", flush=True)
    eng.close()
