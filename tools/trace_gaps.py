"""dev helper: from a rocprofv3 kernel trace CSV, the union busy time of our kernels vs the wall span (idle gaps)"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "wmk::" in r["Kernel_Name"]]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("wmk::")[1].split("<")[0].split("(")[0]) for r in rows)
# skip warmup: take the last 60%
ev = ev[int(len(ev) * 0.4):]
t0, t1 = ev[0][0], max(e[1] for e in ev)
busy, cur_s, cur_e = 0, ev[0][0], ev[0][1]
conc = 0
for s, e, n in ev[1:]:
    if s <= cur_e:
        cur_e = max(cur_e, e)
    else:
        busy += cur_e - cur_s
        cur_s, cur_e = s, e
busy += cur_e - cur_s
tot = sum(e - s for s, e, n in ev)
print(f"span {1e-3*(t1-t0):.1f} us, union busy {1e-3*busy:.1f} us ({100*busy/(t1-t0):.1f}%), sum of kernel durations {1e-3*tot:.1f} us (avg concurrency {tot/busy:.2f})")
per = {}
for s, e, n in ev:
    per.setdefault(n, [0, 0]); per[n][0] += 1; per[n][1] += e - s
for n, (c, d) in sorted(per.items(), key=lambda kv: -kv[1][1]):
    print(f"  {n:18s} {c:5d} launches  avg {1e-3*d/c:8.1f} us  total {1e-3*d:9.1f} us")
