"""Turns the rocprofv3 outputs of tools/run_nvf_profiles.sh (gpurun_out/nvf/) into the committed NVF profiles:
  profiles/<rnd>_nvf_bench_<RxC>.json       the bench line of `bench.py --mask NVF` at that size (3 slots)
  profiles/<rnd>_nvf_kernel_stats.csv       rocprofv3 --kernel-trace --stats rows of the engine's kernels, per size
  profiles/<rnd>_nvf_pmc_summary.json       per size and kernel: counter means, durations, and the derived figures DESIGN.md quotes
                                            (vector instructions per launch, ns per vector instruction and SIMD, HBM bytes)
  profiles/pmc_traffic.json                 + the NVF entries bench.py's roofline.traffic reads (key <RxC>_f32_F<F>_NVF)
HBM bytes = 2 * FETCH_SIZE KiB + WRITE_SIZE KiB (MI355X_MICROARCH.md: FETCH_SIZE reports half the bytes of 16 B/lane reads)."""
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r04"
src = os.path.join(ROOT, "gpurun_out", "nvf")
SIZES = [("1080x1920", 32), ("2160x3840", 16), ("4320x7680", 4)]
NSIMD = 1024
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)

stats_rows, header = [], None
summary = {"_note": "rocprofv3 --kernel-trace --pmc passes of `bench.py --mask NVF --slots 1 --steps 5 --warmup 2 ...` (tools/run_nvf_profiles.sh), one "
                    "counter group per run; means over the launches seen.  Derived: ns_per_valu_inst_per_simd = duration / (SQ_INSTS_VALU / 1024 SIMDs); "
                    "hbm_bytes = 2 * FETCH_SIZE + WRITE_SIZE (KiB -> bytes)."}
traffic_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
traffic = json.load(open(traffic_path)) if os.path.exists(traffic_path) else {}
ALG = {"k_gram": lambda F, N: 4 * F * N, "k_nvf_stats": lambda F, N: (4 * F + 4) * N, "k_embed": lambda F, N: (8 * F + 4) * N, "k_detect": lambda F, N: (4 * F + 4) * N}
for tag, F in SIZES:
    bj = os.path.join(src, f"bench_{tag}.json")
    if not os.path.exists(bj):
        continue
    line = open(bj).read().strip().splitlines()[-1]
    bench = json.loads(line)
    json.dump(bench, open(os.path.join(ROOT, "profiles", f"{rnd}_nvf_bench_{tag}.json"), "w"), indent=1)
    ks = os.path.join(src, f"stats_{tag}", f"{rnd}_kernel_stats.csv")
    if os.path.exists(ks):
        rows = list(csv.reader(open(ks)))
        header = ["size"] + rows[0]
        stats_rows += [[tag] + r for r in rows[1:] if "wmk::" in r[0]]
    out = os.path.join("/tmp", f"nvf_pmc_{tag}.json")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), os.path.join(src, f"pmc_{tag}"), out], stdout=subprocess.DEVNULL)
    d = json.load(open(out))
    R, Cc = (int(v) for v in tag.split("x"))
    N = R * Cc
    ent = {}
    for k, v in d.items():
        if k not in ALG:
            continue
        durs = [v[q] for q in v if q.startswith("dur_us[")]
        dur = sum(durs) / len(durs)
        e = {c: round(val, 1) for c, val in v.items()}
        e["mean_duration_us"] = round(dur, 2)
        e["alg_bytes_per_launch"] = ALG[k](F, N)
        e["frac_of_8TBs"] = round(ALG[k](F, N) / (dur * 1e-6) / 8e12, 4)
        if "SQ_INSTS_VALU" in v:
            e["valu_inst_per_simd"] = round(v["SQ_INSTS_VALU"] / NSIMD, 1)
            e["ns_per_valu_inst_per_simd"] = round(dur * 1e3 / (v["SQ_INSTS_VALU"] / NSIMD), 3)
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            e["hbm_bytes_per_launch"] = int(2 * v["FETCH_SIZE"] * 1024 + v["WRITE_SIZE"] * 1024)
            e["hbm_over_algorithmic"] = round(e["hbm_bytes_per_launch"] / e["alg_bytes_per_launch"], 3)
            traffic.setdefault(f"{R}x{Cc}_f32_F{F}_NVF", {})[k] = {"hbm_bytes_per_launch": e["hbm_bytes_per_launch"]}
        ent[k] = e
    summary[f"{tag}_F{F}"] = ent
if header:
    with open(os.path.join(ROOT, "profiles", f"{rnd}_nvf_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(header)
        w.writerows(stats_rows)
json.dump(summary, open(os.path.join(ROOT, "profiles", f"{rnd}_nvf_pmc_summary.json"), "w"), indent=1)
traffic["captured_nvf"] = rnd
json.dump(traffic, open(traffic_path, "w"), indent=1)
for k, v in summary.items():
    if k.startswith("_"):
        continue
    print(k, {kk: (vv["mean_duration_us"], vv["frac_of_8TBs"], vv.get("ns_per_valu_inst_per_simd"), vv.get("hbm_over_algorithmic")) for kk, vv in v.items()})
