"""dev helper (GPU box): throughput of the BASELINE.json configurations beside the bench line -> gpurun_out/configs.json
(2) 1920x1080 f32, NVF+ME embed+detect   (3) 3840x2160 f32 ME (the bench line)   (4) 3840x2160 u8 Y planes ME
(5) 7680x4320 f32, NVF+ME embed+detect.  Frames resident in HBM, 3 slots, batched launches; algorithmic bytes per
SURVEY.md section 8d (embed-ME 24N, detect-ME 12N, embed-NVF 20N, detect-NVF 12N; u8 frames: 12N + 6N)."""
import io
import json
import os
import re
import sys
from contextlib import redirect_stdout

import torch

sys.path.insert(0, "tools")
from quick_bench import run

CASES = [
    ("configs[1] 1920x1080 f32 ME", 1080, 1920, 64, 12, torch.float32, 0),   # (64 x 8.3 MB = the bytes of a 16-frame 4K launch)
    ("configs[1] 1920x1080 f32 NVF", 1080, 1920, 64, 12, torch.float32, 1),
    ("configs[2] 3840x2160 f32 ME", 2160, 3840, 16, 20, torch.float32, 0),
    ("3840x2160 f32 NVF", 2160, 3840, 16, 20, torch.float32, 1),
    ("configs[3] 3840x2160 u8 ME", 2160, 3840, 16, 20, torch.uint8, 0),
    ("configs[4] 7680x4320 f32 ME", 4320, 7680, 8, 8, torch.float32, 0),
    ("configs[4] 7680x4320 f32 NVF", 4320, 7680, 8, 8, torch.float32, 1),
]
out = []
for name, R, C, F, iters, dt, mask in CASES:
    buf = io.StringIO()
    with redirect_stdout(buf):
        run(R, C, F, 3, iters, dtype=dt, mask=mask)
    line = buf.getvalue()
    fps = float(re.search(r"([\d.]+) frames/s", line).group(1))
    es = 4 if dt == torch.float32 else 1
    N = R * C
    # SURVEY.md 8d's unit counts W in every sweep of every frame; HBM has to deliver W once per launch of F frames (L2
    # serves the other uses): hbm_bytes_per_frame = frame planes once per sweep + W / F per sweep that reads it
    per_frame = ((es) + (es + 4) + (2 * es + 4) + (es) + (es + 4)) * N if mask == 0 else ((es + 4) + (2 * es + 4) + (es) + (es + 4)) * N
    hbm = (6 * es + 3 * 4.0 / F) * N if mask == 0 else (5 * es + 3 * 4.0 / F) * N
    # the same loop with the opt-in Gram hand-over (detector on WM_MEM_SLOT_OUT; f32 frames only): reported beside, DESIGN 3c
    fps_ho = None
    if dt == torch.float32:
        os.environ["WM_QB_HANDOVER"] = "1"
        buf = io.StringIO()
        with redirect_stdout(buf):
            run(R, C, F, 3, iters, dtype=dt, mask=mask)
        os.environ.pop("WM_QB_HANDOVER")
        fps_ho = float(re.search(r"([\d.]+) frames/s", buf.getvalue()).group(1))
    out.append({"config": name, "frames_per_launch": F, "slots": 3, "frames_per_s": fps, "hbm_bytes_per_frame": int(hbm),
                "achieved_GBs": round(fps * hbm / 1e9, 1), "frac_of_hbm_peak": round(fps * hbm / 8e12, 4),
                "survey_unit_bytes_per_frame": per_frame, "survey_unit_GBs": round(fps * per_frame / 1e9, 1),
                "frames_per_s_with_gram_hand_over": fps_ho})
    print(json.dumps(out[-1]), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/configs.json", "w"), indent=1)
