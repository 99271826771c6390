"""Reads a rocprofv3 kernel-trace CSV: per queue, the qp_admm_rl_kernel launches; prints how much of the time two or more
ADMM kernels were resident at once."""
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "qp_admm_rl_kernel<5" in r["Kernel_Name"]]
ev = []
qs = {}
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    ev.append((s, 1)); ev.append((e, -1))
    qs.setdefault(r["Queue_Id"], []).append((s, e))
ev.sort()
lvl = 0; last = ev[0][0]; tot = {}
for t, d in ev:
    tot[lvl] = tot.get(lvl, 0) + (t - last); last = t; lvl += d
print("launches", len(rows), "queues", {q: len(v) for q, v in qs.items()})
print("ms with k ADMM kernels resident:", {k: round(v / 1e6, 1) for k, v in sorted(tot.items())})
for q, v in qs.items():
    print("queue", q, "sum of durations %.1f ms" % (sum(e - s for s, e in v) / 1e6), "first", [round((e - s) / 1e6, 2) for s, e in v[:12]])
# timeline of the last step: launches after the last gap of more than 5 ms with no kernel... simply the last third
allr = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"]) for r in rows))
n3 = len(allr) // 3
last = allr[2 * n3:]
t0 = last[0][0]
for s, e, q in last:
    print("q%s start %8.2f dur %6.2f" % (q, (s - t0) / 1e6, (e - s) / 1e6))
print("last step span %.1f ms" % ((max(e for s, e, q in last) - t0) / 1e6))
