"""The bench line's contract, checked on the committed line of the last profile run (profiles/r02_bench_n1.json is what
`python bench.py` printed on the GPU box): the keys the driver parses, the roofline / cpu_baseline objects, and the
arithmetic that ties them together.  No GPU needed."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def line():
    with open(os.path.join(ROOT, "profiles", "r02_bench_n1.json")) as f:
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
    assert line["parity"]["max_abs_dcorr_vs_oracle"] <= line["parity"]["tolerance"]["corr_abs"]
