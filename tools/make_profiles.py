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
rnd = sys.argv[1] if len(sys.argv) > 1 else "r03"
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
# per-LAUNCH trace of the same run (not only --stats): every kernel's launches in start order, k_gram split into its two
# populations -- the embed side reads x (the first k_gram of a step), the detect side reads y, which k_embed has just
# written with non-temporal stores (the second) -- so that a bimodal average can be attributed
import glob  # noqa: E402
import statistics  # noqa: E402
kt = glob.glob(os.path.join(ROOT, "gpurun_out", "prof_stats", "**", f"{rnd}_kernel_trace.csv"), recursive=True)
if kt:
    launches = []
    for row in csv.DictReader(open(kt[0])):
        if "wmk::" in row["Kernel_Name"]:
            body = row["Kernel_Name"].split("wmk::")[1]
            kname = body.split("<")[0].split("(")[0]
            if kname == "k_embed" and "<" in body:
                targs = [a.strip() for a in body.split("<", 1)[1].split(">")[0].split(",")]
                if len(targs) >= 8 and targs[7] in ("true", "1"):
                    kname = "k_embed[hand-over]"   # (the path_slot_out leg: HO = true)
            launches.append((int(row["Start_Timestamp"]), kname, (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3))
    launches.sort()
    groups, side = {}, 0
    for t0, name, us in launches:
        if name == "k_gram":
            gkey = "k_gram[embed side: reads x]" if side == 0 else "k_gram[detect side: reads y]"
            side ^= 1
        else:
            gkey = name
            if name in ("k_me_stats", "k_nvf_stats"):
                side = 1   # (re-synchronise: the k_gram after a stats sweep is the detect side's)
            if name == "k_detect":
                side = 0
            if name == "k_embed[hand-over]":
                side = 0   # (no k_gram on the detect side of the hand-over leg: the next k_gram is an embed's)
        groups.setdefault(gkey, []).append(us)
    summ = {k: {"launches": len(v), "mean_us": round(statistics.mean(v), 2), "sd_us": round(statistics.pstdev(v), 2), "min_us": round(min(v), 2),
                "max_us": round(max(v), 2), "median_us": round(statistics.median(v), 2)} for k, v in groups.items()}
    summ["_note"] = ("rocprofv3 --kernel-trace of `bench.py --steps 5 --warmup 2 --slots 1 --frames-per-slot 16 ...`: warm-up + timed steps (one slot: "
                     "embed and detect of a batch back to back) + the 10 hipEvent steps (every launch bracketed by events, host sync per step)")
    json.dump(summ, open(os.path.join(ROOT, "profiles", f"{rnd}_kernel_trace_summary.json"), "w"), indent=1)
    print(json.dumps(summ, indent=1)[:2000])
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
    # fused kernels: wide row loads (16 B per lane: x / y rows, W rows) are under-counted by half, the narrow ones (W at the halo
    # column: one dword per row and wave; the halo-pair gather: 8 B per lane, one line per row; border chunks; sc1 record
    # loads) are not (MI355X_MICROARCH.md: "other access widths are uncalibrated").  Doubling everything therefore over-counts
    # the narrow share once.  Model at 3840x2160 f32, 255 workgroups x 16 waves: narrow lines = waves x 10 rows x 2 sides x 64 B for
    # the W halo column (5.2 MB) + the same for the halo-pair gather (5.2 MB) ~ 10.4 MB; reads = 2 * FETCH - narrow
    NARROW_4K = 2 * 255 * 16 * 10 * 2 * 64
    for k, v in d1.items():
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            v["hbm_bytes_per_launch"] = int(v["FETCH_SIZE"] * 1024 * 2 + v["WRITE_SIZE"] * 1024)
            v["hbm_bytes_per_launch_narrow_reads_counted_once"] = int(v["FETCH_SIZE"] * 1024 * 2 - NARROW_4K + v["WRITE_SIZE"] * 1024)
            v["note"] = ("2*FETCH_SIZE + WRITE_SIZE (gfx950 correction for wide reads); the second figure takes the modelled narrow reads "
                         "(W halo column + halo-pair gather, ~10.4 MB at 4K) out of the doubling; per mask, tools/f1_trace.py")
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
