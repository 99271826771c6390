"""rocprofv3 kernel-trace CSV -> durations of the ADMM launches of the last step, in launch order."""
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "qp_admm_rl_kernel<5" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows)
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
print("launches", len(rows))
print(" ".join("%.1f" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6) for r in rows))
print("span %.1f ms, sum %.1f ms" % ((int(rows[-1]["End_Timestamp"]) - t0) / 1e6, sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows) / 1e6))
