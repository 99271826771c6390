"""Summarise rocprofv3 --pmc passes (per kernel: sum over dispatches).

gfx950 corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts 64 B per
128-B request for wide coalesced reads, so it is doubled; WRITE_SIZE is exact for
16-B-per-lane streaming stores.  Both counters are in KiB."""
import csv, glob, os, sys, json, collections
root = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:44]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
out = {}
for k in sorted(tot, key=lambda k: -tot[k].get("SQ_WAVE_CYCLES", tot[k].get("FETCH_SIZE", 0))):
    print(k)
    out[k] = {}
    for c, v in sorted(tot[k].items()):
        n = cnt[k][c]
        extra = ""
        if c == "FETCH_SIZE": extra = "  -> %.3f MiB corrected (x2), %.4f MiB/launch" % (2 * v / 1024, 2 * v / 1024 / n)
        if c == "WRITE_SIZE": extra = "  -> %.3f MiB, %.4f MiB/launch" % (v / 1024, v / 1024 / n)
        print("   %-24s %18.0f  (%d dispatches)%s" % (c, v, n, extra))
        out[k][c] = {"sum": v, "dispatches": n}
json.dump(out, open(os.path.join(root, "pmc_summary.json"), "w"), indent=1)
