"""Turns the rocprofv3 outputs of one GPU session (gpurun_out/) into the committed profiles/ summaries:
  profiles/r01_kernel_stats.csv        rocprofv3 --kernel-trace --stats summary (our kernels + totals)
  profiles/r01_pmc_summary.json        per-kernel means of the --pmc passes
  profiles/pmc_traffic.json            HBM bytes per launch per kernel (read by bench.py's roofline.traffic)
HBM bytes = FETCH_SIZE * 1024 * 2 + WRITE_SIZE * 1024: on gfx950 FETCH_SIZE reports half the bytes of wide
(16 B/lane) coalesced reads (MI355X_MICROARCH.md, HBM section), WRITE_SIZE is exact for 16 B/lane stores."""
import csv, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
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
pm = os.path.join(ROOT, "gpurun_out", "pmc")
if os.path.isdir(pm):
    out = os.path.join(ROOT, "profiles", f"{rnd}_pmc_summary.json")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), pm, out], stdout=subprocess.DEVNULL)
    d = json.load(open(out))
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    tj = json.load(open(tpath)) if os.path.exists(tpath) else {}
    tj[key] = {}
    for k, v in d.items():
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            tj[key][k] = {"hbm_bytes_per_launch": int(v["FETCH_SIZE"] * 1024 * 2 + v["WRITE_SIZE"] * 1024),
                          "FETCH_SIZE_KB_raw": v["FETCH_SIZE"], "WRITE_SIZE_KB_raw": v["WRITE_SIZE"],
                          "correction": "FETCH_SIZE x2 (gfx950 wide-read under-count), WRITE_SIZE exact",
                          "dur_us": v.get("dur_us[fetch]")}
    json.dump(tj, open(tpath, "w"), indent=1)
    print(json.dumps(tj[key], indent=1)[:1500])
