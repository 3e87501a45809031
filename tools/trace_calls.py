"""dev helper: timeline of the last N kernel launches in a rocprofv3 kernel trace (start offsets, durations, gaps)"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "wmk::" in r["Kernel_Name"] or "rocclr" in r["Kernel_Name"]]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("wmk::")[-1].split("<")[0].split("(")[0]) for r in rows)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
ev = ev[-n:]
t0 = ev[0][0]; prev_end = None
for s, e, name in ev:
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f} us  gap {gap:6.1f}  {name}")
    prev_end = e
