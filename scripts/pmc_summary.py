"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes for the ADMM kernel.

gfx950 corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts 64 B per
128-B request for wide coalesced reads, so it is doubled; WRITE_SIZE is exact for
16-B-per-lane streaming stores.  Both counters are in KiB."""
import csv, glob, os, sys, json, collections
root = sys.argv[1]
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(os.path.join(root, "pmc_" + c, "**", "*counter_collection.csv"), recursive=True)
    tot = collections.defaultdict(float); cnt = collections.defaultdict(int)
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == c:
                k = r["Kernel_Name"].split("(")[0][:40]
                tot[k] += float(r["Counter_Value"]); cnt[k] += 1
    out[c] = {k: (tot[k], cnt[k]) for k in tot}
for c, d in out.items():
    for k, (v, n) in sorted(d.items(), key=lambda kv: -kv[1][0])[:6]:
        kib = v * (2 if c == "FETCH_SIZE" else 1)
        print("%-11s %-42s launches %4d  total %.3f MiB (corrected)  per launch %.3f MiB" % (c, k, n, kib / 1024, kib / 1024 / max(n, 1)))
json.dump({c: {k: {"raw_KiB": v, "launches": n} for k, (v, n) in d.items()} for c, d in out.items()},
          open(os.path.join(root, "pmc_summary.json"), "w"), indent=1)
