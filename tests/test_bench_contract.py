"""The bench line's contract, checked on the committed line of the last profile run (profiles/r04_bench_n1.json is what
`python bench.py` printed on the GPU box): the keys the driver parses, the roofline / cpu_baseline objects, and the
arithmetic that ties them together.  No GPU needed."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def line():
    with open(os.path.join(ROOT, "profiles", "r04_bench_n1.json")) as f:
        return json.load(f)


def test_top_level_keys_and_types(line):
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["unit"] == "frames/s" and line["higher_is_better"] is True and line["scaling"] == "weak"
    assert line["vs_baseline"] is None  # BASELINE.md publishes no number for this metric
    assert line["n_gpus"] == 1 and line["dtype"] == "f32" and line["data"] == "synthetic"
    assert "workload" in line["config"] and "model" not in line["config"]
    # value = frames of the timed steps / time
    fps = line["config"]["frames_per_step_per_gpu"] * line["n_gpus"] / (line["ms_per_step"] * 1e-3)
    assert abs(fps - line["value"]) <= 2e-3 * line["value"]


def test_roofline_object(line):
    r = line["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["peak"] == 8000.0  # MI355X_MICROARCH.md: HBM3E ~8 TB/s
    # achieved = algorithmic bytes per launch / the kernel's average launch duration
    assert abs(r["achieved"] - r["alg_bytes_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e9) <= 1e-3 * r["achieved"]
    assert r["traffic"] is None or r["traffic"] >= r["alg_bytes_per_launch"]  # PMC bytes: never below the algorithmic ones
    # the path figure counts W once per launch, and stays below what HBM can deliver
    assert line["path"]["frac_of_hbm_peak"] < 0.8 and line["path"]["hbm_bytes_per_frame"] < line["path"]["survey_unit"]["bytes_per_frame"]


def test_cpu_baseline_and_single_call_objects(line):
    c = line["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0
    s = line["single_call"]
    assert s["path"].startswith("fused") and abs(s["us_per_frame"] - (s["embed_us"] + s["detect_us"])) < 0.05 * s["us_per_frame"]
    assert s["same_calls_on_the_sweeps"]["us_per_frame"] > s["us_per_frame"]
    # the lead fraction of the one-image-per-call figure counts the bytes the fused kernels have to move (20 N at f32), not the sweeps' 36 N
    assert "compulsory" in s["frac_definition"] and s["frac"] < s["frac_in_the_sweeps_unit_36N"] and s["frac"] < 0.5
    assert line["parity"]["max_abs_dcorr_vs_oracle"] <= line["parity"]["tolerance"]["corr_abs"]


def test_multi_gpu_record_and_sustained_leg(line):
    """what answers "did the communicator see N ranks" and "does the rate hold for a second" from the record alone"""
    assert line["ranks_seen"] == line["n_gpus"] == len(line["per_rank_frames_per_s"])
    sus = line["sustained"]
    assert sus["seconds"] >= 1.0 and abs(sus["frames_per_s"] - line["value"]) < 0.1 * line["value"]
    # the per-kernel event durations are cross-checked against the wall time of the same serial steps
    chk = line["serial_step_check"]
    assert abs(chk["sum_of_kernel_event_us"] - chk["wall_us_per_step"]) < 0.12 * chk["wall_us_per_step"]


def test_gram_hand_over_leg_is_reported_beside_the_headline(line):
    """the opt-in hand-over (wm_set_handover + WM_MEM_SLOT_OUT) has a leg of its own: it never replaces `value`"""
    ho = line["path_slot_out"]
    assert ho["seconds"] >= 0.3 and "k_gram_ho" in ho["kernels_avg_us"]
    assert 1.0 < ho["vs_independent_calls"] < 1.3 and ho["frames_per_s"] > line["value"]
    assert ho["max_abs_score_difference_to_independent_calls"] <= 2e-7
    # the headline's own kernels do not include the hand-over's
    assert "k_gram_ho" not in line["kernels"]


def test_placement_box_record_and_memory_yardstick(line):
    """round 4: where the rank ran (NUMA node of its GPU, confirmed by the runtime's PCI address), what the box looked like
    (partition modes, memory clock, versions) and what a pure store / copy / read kernel reaches on it -- the records that let two
    boxes' different k_embed times be read off the line"""
    pl = line["placement"]
    assert len(pl) == line["n_gpus"]
    for k in ("numa_node", "cpus", "applied", "confirmed_by_pci"):
        assert k in pl[0], k
    assert pl[0]["applied"] is False  # a one-rank run is described, not pinned
    box = line["box"]
    for k in ("current_memory_partition", "current_compute_partition", "mem_info_vram_used", "mclk", "vbios_version", "rocm", "firmware"):
        assert k in box, k
    mb = line["membench"]
    assert mb["bytes_per_launch"] == line["kernels"]["k_gram"]["alg_bytes_per_launch"]  # the bytes of one 16-frame launch's planes
    for name in ("store", "copy", "read", "store_best_shape", "copy_best_shape", "read_best_shape"):
        assert 1000.0 < mb[name]["GBs"] < 8000.0, (name, mb[name])
    # the best streaming shape is at least as fast as the sweep-like grid, and no kernel of the path beats a pure read of its bytes
    assert mb["read_best_shape"]["GBs"] >= 0.95 * mb["read"]["GBs"]
    for k, v in line["kernels"].items():
        if k != "k_embed" and "achieved_GBs" in v:
            assert v["achieved_GBs"] <= 1.02 * mb["read_best_shape"]["GBs"], k
    st = line["stream"]
    assert len(st["host_staged_GBs_each_way_by_rank"]) == line["n_gpus"] and st["pinned_ring_MB_per_rank"] > 0


# ---- the launcher-free multi-rank entry (`python bench.py --gpus N`, no torch.distributed.run around it) ------------------
def _bench(*argv, env=None):
    import subprocess
    import sys
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=e, capture_output=True, text=True, timeout=300)


@pytest.mark.parametrize("n", [2, 3])
def test_gpus_n_without_a_launcher_spawns_the_ranks(n):
    """rank processes started by bench.py itself rendezvous (gloo here), count each other, gather scores into stream order, and
    the parent's stdout is exactly rank 0's one JSON line"""
    p = _bench("--gpus", str(n), "--plumbing-only")
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == n and rec["ranks_seen"] == n and rec["scores_in_stream_order"] is True


def test_a_failing_rank_fails_the_launcher():
    """no GPU here: every rank dies on the 'needs a HIP device' assertion; the launcher must exit non-zero and print no line"""
    p = _bench("--gpus", "2", "--steps", "1", "--warmup", "0")
    assert p.returncode != 0
    assert "needs a HIP device" in p.stderr and "exited with code" in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]


def test_the_launcher_process_never_imports_torch():
    """the parent of a launcher-free run must not have touched torch (let alone HIP) when it starts the ranks"""
    import subprocess
    import sys
    code = ("import sys, runpy\nsys.argv = ['bench.py', '--gpus', '2', '--plumbing-only']\n"
            "try:\n    runpy.run_path(%r, run_name='__main__')\nexcept SystemExit as e:\n    rc = e.code\n"
            "print('TORCH_IN_PARENT', 'torch' in sys.modules, 'RC', rc)\n" % os.path.join(ROOT, "bench.py"))
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    p = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=300)
    assert "TORCH_IN_PARENT False RC 0" in p.stdout, p.stdout + p.stderr[-1500:]


def test_launcher_form_still_works_under_torchrun_env():
    """with RANK / WORLD_SIZE in the environment (the driver's torch.distributed.run form) bench.py does not spawn"""
    p = _bench("--gpus", "1", "--plumbing-only", env={"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1"})
    assert p.returncode == 0 and json.loads(p.stdout.strip().splitlines()[-1])["n_gpus"] == 1
