#!/bin/bash
# dev helper (GPU box): the reference's single-image benchmark protocol (testForImage) at 3840x2160 through wm_app
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
python3 - <<'PY'
import sys, importlib, numpy as np
sys.path.insert(0, '.')
s = importlib.import_module("watermarking-gpu_amd.synth")
Rr, C = 2160, 3840
rgb = np.stack([s.synth_frames_torch(Rr, C, 1, "cuda", dtype="u8", first_frame=k)[0].cpu().numpy() for k in range(3)], axis=-1)
open("/tmp/4k.ppm", "wb").write(b"P6\n%d %d\n255\n" % (C, Rr) + rgb.tobytes())
PY
./watermarking-gpu_amd/wm_genw 2160 3840 28390211 /tmp/w_4k.dat
cat > /tmp/settings_4k.ini <<INI
[paths]
image = /tmp/4k.ppm
watermark = /tmp/w_4k.dat
[options]
opencl_device = 0
execution_time_in_fps = true
[parameters]
p = 3
psnr = 40.0
loops_for_test = ${LOOPS:-200}
INI
./watermarking-gpu_amd/wm_app /tmp/settings_4k.ini
