"""dev helper: per-kernel means of rocprofv3 --pmc passes (gpurun_out/pmc/*/ *_counter_collection.csv)
joined with kernel durations from the same pass's kernel trace -> prints a table and writes JSON."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"


def short_name(name):
    """wmk::k_x<...>(...) -> k_x; the fused kernels keep their mask ([ME] / [NVF]: template argument MASK), since the two masks
    move different bytes and must not be averaged together"""
    body = name.split("wmk::")[1]
    base = body.split("(")[0].split("<")[0]
    if base in ("k_fused_embed", "k_fused_detect") and "<" in body:
        targs = [a.strip() for a in body.split("<", 1)[1].split(">")[0].split(",")]
        mask = targs[3] if base == "k_fused_embed" else targs[1]
        base += "[ME]" if mask == "0" else "[NVF]"
    if base == "k_embed" and "<" in body:
        # the Gram hand-over instantiation (last template argument HO = true) does more per row: its own row
        targs = [a.strip() for a in body.split("<", 1)[1].split(">")[0].split(",")]
        if len(targs) >= 8 and targs[7] in ("true", "1"):
            base += "[hand-over]"
    return base


out = defaultdict(lambda: defaultdict(list))
dur = defaultdict(list)
for d in sorted(glob.glob(os.path.join(root, "*"))):
    cc = glob.glob(os.path.join(d, "*_counter_collection.csv"))
    kt = glob.glob(os.path.join(d, "*_kernel_trace.csv"))
    if not cc:
        continue
    for row in csv.DictReader(open(cc[0])):
        name = row["Kernel_Name"]
        if "wmk::" not in name:
            continue
        short = short_name(name)
        out[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
    if kt:
        for row in csv.DictReader(open(kt[0])):
            name = row["Kernel_Name"]
            if "wmk::" not in name:
                continue
            short = short_name(name)
            dur[(os.path.basename(d), short)].append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
res = {}
for k, cs in out.items():
    res[k] = {c: sum(v) / len(v) for c, v in cs.items()}
    res[k]["launches_seen"] = max(len(v) for v in cs.values())
for (p, k), v in dur.items():
    res.setdefault(k, {})[f"dur_us[{p}]"] = sum(v) / len(v) / 1e3
print(json.dumps(res, indent=1))
if len(sys.argv) > 2:
    json.dump(res, open(sys.argv[2], "w"), indent=1)
