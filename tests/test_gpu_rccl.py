"""RCCL on the box's GPU before the multi-GPU driver run needs it: a fresh child process initialises the "nccl"
(= RCCL) backend with world_size 1 AFTER the engine's library is loaded, pushes a batch through the engine, gathers the
per-frame detector scores with frames.gather_scores on a DEVICE tensor (async all_gather_into_tensor on RCCL's stream)
and compares them with the scores of a plain single-process run; it also checks that the process holds exactly ONE HIP
runtime (PyTorch ships its own libamdhip64: libwm_hip.so must bind to that one, not bring a second)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import importlib, json, os, sys
import numpy as np
sys.path.insert(0, os.environ["WM_ROOT"]); sys.path.insert(0, os.path.join(os.environ["WM_ROOT"], "tests"))
import torch
import torch.distributed as dist
wm = importlib.import_module("watermarking-gpu_amd")          # the engine's library first ...
frames = importlib.import_module("watermarking-gpu_amd.frames")
from synth import synth_frame, synth_watermark
wm.lib()
use_dist = os.environ["WM_CHILD_DIST"] == "1"
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
if use_dist:                                                     # ... then RCCL
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
R, C, F = 270, 512, 6
W = synth_watermark(R, C)
xs = torch.from_numpy(np.stack([synth_frame(R, C, frame=f) for f in range(F)])).cuda()
eng = wm.Watermark(R, C, W, 3, 40.0, nslots=1, max_frames=F)
ys, a = eng.makeWatermark(xs, xs, wm.MASK_TYPE.ME)
corr = eng.detectWatermark(ys, wm.MASK_TYPE.ME)
scores = torch.tensor(corr, dtype=torch.float32, device=dev)
if use_dist:
    (recv, finish), work = frames.gather_scores(scores, F, 0, 1, device=dev, async_op=True, force_collective=True)
    work.wait()
    out = finish().cpu().numpy()
    t = torch.ones(8, device=dev)
    dist.all_reduce(t)                                          # one more collective on the same communicator
    assert float(t.sum()) == 8.0
else:
    out = scores.cpu().numpy()
maps = open("/proc/self/maps").read()
hips = sorted({l.split()[-1] for l in maps.splitlines() if "libamdhip64" in l})
rccl = sorted({l.split()[-1] for l in maps.splitlines() if "librccl" in l})
print("RESULT " + json.dumps({"scores": [float(v) for v in out], "a": [float(v) for v in a], "hip_runtimes": hips, "rccl": rccl}))
eng.close()
if use_dist:
    dist.destroy_process_group()
"""


def run_child(with_dist, port):
    env = dict(os.environ, WM_ROOT=ROOT, WM_CHILD_DIST="1" if with_dist else "0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[7:])


def test_rccl_world1_after_engine_load():
    plain = run_child(False, 29641)
    rccl = run_child(True, 29642)
    assert len(rccl["hip_runtimes"]) == 1, rccl["hip_runtimes"]
    assert len(plain["hip_runtimes"]) == 1, plain["hip_runtimes"]
    assert rccl["rccl"], "the RCCL library was not loaded: the nccl backend did not run"
    np.testing.assert_array_equal(np.array(rccl["scores"], np.float32), np.array(plain["scores"], np.float32))
    np.testing.assert_array_equal(np.array(rccl["a"], np.float32), np.array(plain["a"], np.float32))
    assert all(0.2 < s < 1.0 for s in rccl["scores"])


def test_bench_two_ranks_without_a_launcher_on_one_gpu():
    """`python bench.py --gpus 2` with NO launcher around it: bench.py starts the two ranks itself (before touching the GPU),
    both share this box's one GPU (WM_BENCH_ALL_ON_DEVICE0, gloo instead of RCCL: two RCCL ranks cannot share a device), and
    rank 0's single JSON line comes back with n_gpus 2, both ranks counted and a rate per rank"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(WM_BENCH_ALL_ON_DEVICE0="1", WM_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--rows", "1080", "--cols", "1920",
                        "--frames-per-slot", "4", "--slots", "2", "--sustain-seconds", "0.2"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-1000:] + p.stderr[-4000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["ranks_seen"] == 2 and rec["backend"] == "gloo"
    assert len(rec["per_rank_frames_per_s"]) == 2 and all(v > 0 for v in rec["per_rank_frames_per_s"])
    assert rec["value"] > 0 and rec["sustained"]["frames_per_s"] > 0 and rec["sustained"]["seconds"] >= 0.2
    assert "cpu_baseline" not in rec  # rank 0 at N = 1 only
    # every rank's host placement is in the line (gathered over the process group): both ranks sit on the same GPU here
    assert len(rec["placement"]) == 2 and all("numa_node" in p_ for p_ in rec["placement"])
    assert rec["placement"][0]["pci_of_open_device"] == rec["placement"][1]["pci_of_open_device"]


def test_bench_nvf_mask_line():
    """`bench.py --mask NVF` (BASELINE.json configs[1] / [4] name the NVF mask): four sweeps per frame, the NVF kernels in the
    `kernels` block with their fractions, no ME-only legs"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--mask", "NVF", "--steps", "3", "--warmup", "1", "--rows", "1080", "--cols", "1920",
                        "--frames-per-slot", "8", "--slots", "2", "--sustain-seconds", "0.2", "--cpu-seconds", "1"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-1000:] + p.stderr[-4000:]
    rec = json.loads([l for l in p.stdout.splitlines() if l.strip()][-1])
    assert "NVF mask" in rec["metric"] and "k_nvf_stats" in rec["kernels"] and "k_me_stats" not in rec["kernels"]
    assert rec["kernels"]["k_gram"]["launches"] == rec["kernels"]["k_detect"]["launches"]  # one Gram sweep per frame: the detector's
    assert rec["path"]["definition"].startswith("4 sweeps")
    assert "stream" not in rec and "single_call" not in rec
    assert rec["parity"]["max_abs_dcorr_vs_oracle"] <= 1e-5 and rec["parity"]["max_rel_da_vs_oracle"] <= 1e-4
    assert rec["path_slot_out"]["frames_per_s"] > 0
