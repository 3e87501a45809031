"""dev helper: from a rocprofv3 kernel trace CSV of tools/f1_trace.py: per kernel the average duration and the average
idle gap before it (end of the previous kernel to its start)"""
import csv
import sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "wmk::" in r["Kernel_Name"]]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("wmk::")[1].split("<")[0].split("(")[0]) for r in rows)
ev = ev[len(ev) // 3:]
per = {}
for (s0, e0, n0), (s1, e1, n1) in zip(ev, ev[1:]):
    d = per.setdefault(n1, [0, 0, 0, {}])
    d[0] += 1; d[1] += e1 - s1; d[2] += s1 - e0
    d[3][n0] = d[3].get(n0, 0) + 1
for n, (c, dur, gap, prev) in per.items():
    print(f"{n:16s} {c:5d}  avg {1e-3*dur/c:7.2f} us   gap before {1e-3*gap/c:7.2f} us   after {max(prev, key=prev.get)}")
