"""Turns the rocprofv3 outputs of one GPU session (gpurun_out/) into the committed profiles/ summaries:
  profiles/<rnd>_kernel_stats.csv      rocprofv3 --kernel-trace --stats summary (our kernels + totals) of the serialised bench
  profiles/<rnd>_pmc_summary.json      per-kernel means of the --pmc passes
  profiles/<rnd>_single_call_*         the same for the one-image-per-call loop (fused kernels)
  profiles/<rnd>_bench_n1*.json        the bench lines of the session
  profiles/pmc_traffic.json            HBM bytes per launch per kernel (read by bench.py's roofline.traffic)
HBM bytes = FETCH_SIZE * 1024 * 2 + WRITE_SIZE * 1024: on gfx950 FETCH_SIZE reports half the bytes of wide
(16 B/lane) coalesced reads (MI355X_MICROARCH.md, HBM section), WRITE_SIZE is exact for 16 B/lane stores."""
import csv, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r02"
key = sys.argv[2] if len(sys.argv) > 2 else "2160x3840_f32_F16"
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
ks = os.path.join(ROOT, "gpurun_out", "prof_stats", f"{rnd}_kernel_stats.csv")
if os.path.exists(ks):
    rows = list(csv.reader(open(ks)))
    with open(os.path.join(ROOT, "profiles", f"{rnd}_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(rows[0])
        for r in rows[1:]:
            if "wmk::" in r[0] or "rocclr" in r[0]:
                w.writerow(r)
# the one-image-per-call loop (fused kernels; tools/f1_trace.py: ME then NVF, embed / detect / pairs)
ks1 = os.path.join(ROOT, "gpurun_out", "prof_stats_single", f"{rnd}_single_kernel_stats.csv")
if os.path.exists(ks1):
    rows = list(csv.reader(open(ks1)))
    with open(os.path.join(ROOT, "profiles", f"{rnd}_single_call_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(rows[0])
        for r in rows[1:]:
            if "wmk::" in r[0]:
                w.writerow(r)
pm1 = os.path.join(ROOT, "gpurun_out", "pmc_single")
if os.path.isdir(pm1):
    out1 = os.path.join(ROOT, "profiles", f"{rnd}_single_call_pmc_summary.json")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), pm1, out1], stdout=subprocess.DEVNULL)
    d1 = json.load(open(out1))
    for k, v in d1.items():
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            v["hbm_bytes_per_launch"] = int(v["FETCH_SIZE"] * 1024 * 2 + v["WRITE_SIZE"] * 1024)
            v["note"] = "2*FETCH_SIZE + WRITE_SIZE (gfx950 correction); means over the launches of tools/f1_trace.py (ME and NVF calls mixed)"
    json.dump(d1, open(out1, "w"), indent=1)
    print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk in ("hbm_bytes_per_launch", "dur_us[fetch]", "SQ_INSTS_VALU")} for k, v in d1.items()}, indent=1))
for name in (f"bench_{rnd}.json", f"bench_{rnd}_u8.json"):
    src = os.path.join(ROOT, "gpurun_out", name)
    if os.path.exists(src):
        line = open(src).read().strip().splitlines()[-1]
        json.loads(line)
        open(os.path.join(ROOT, "profiles", name.replace("bench_" + rnd, rnd + "_bench_n1")), "w").write(line + "\n")
pm = os.path.join(ROOT, "gpurun_out", "pmc")
if os.path.isdir(pm):
    out = os.path.join(ROOT, "profiles", f"{rnd}_pmc_summary.json")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), pm, out], stdout=subprocess.DEVNULL)
    d = json.load(open(out))
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    tj = json.load(open(tpath)) if os.path.exists(tpath) else {}
    tj[key] = {}
    tj["captured"] = rnd
    for k, v in d.items():
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            tj[key][k] = {"hbm_bytes_per_launch": int(v["FETCH_SIZE"] * 1024 * 2 + v["WRITE_SIZE"] * 1024),
                          "FETCH_SIZE_KB_raw": v["FETCH_SIZE"], "WRITE_SIZE_KB_raw": v["WRITE_SIZE"],
                          "correction": "FETCH_SIZE x2 (gfx950 wide-read under-count), WRITE_SIZE exact",
                          "dur_us": v.get("dur_us[fetch]")}
    json.dump(tj, open(tpath, "w"), indent=1)
    print(json.dumps(tj[key], indent=1)[:1500])
