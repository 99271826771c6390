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

# ---- traffic record of the dominant ADMM kernel, keyed by the kernel sources it was measured on (bench.py quotes it
# only while sco_py_amd/csrc is unchanged): FETCH_SIZE (KiB) x 2 (gfx950 counts 64 B per 128-B request on wide streaming
# reads, MI355X_MICROARCH.md HBM section) + WRITE_SIZE (KiB), one bench step (--steps 1 --warmup 0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
for key, fname in (("qp_admm_rl_kernel", "traffic.json"), ("qp_admm_bt_kernel", "traffic_12x50.json")):
    ks = [k for k in out if key in k and "FETCH_SIZE" in out[k] and "WRITE_SIZE" in out[k]]
    if not ks:
        continue
    fetch = sum(out[k]["FETCH_SIZE"]["sum"] for k in ks); write = sum(out[k]["WRITE_SIZE"]["sum"] for k in ks)
    launches = sum(out[k]["FETCH_SIZE"]["dispatches"] for k in ks)
    rec = {"kernel": ", ".join(ks), "kernel_src_sha": bench.kernel_src_sha(),
           "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, bench.py --steps 1 --warmup 0 (scripts/gpu_pmc.sh)",
           "batch": int(os.environ.get("PMC_BATCH", "1024")),
           "fetch_KiB_raw": fetch, "write_KiB_raw": write, "launches": launches,
           "correction": "FETCH_SIZE x2 (gfx950 counts 64 B per 128-B request), WRITE_SIZE x1",
           "hbm_bytes_per_step": (2 * fetch + write) * 1024.0, "hbm_bytes_per_launch": (2 * fetch + write) * 1024.0 / max(launches, 1)}
    if "SQ_INSTS_VALU" in out[ks[0]]:
        rec["sq"] = {c: sum(out[k][c]["sum"] for k in ks if c in out[k]) for c in
                     ("SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU",
                      "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
                      "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "GRBM_GUI_ACTIVE")}
    try:        # shader clock held during the kernel: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / kernel time of the same pass
        dur = 0
        for f in glob.glob(os.path.join(root, "pmc_sq2", "**", "*kernel_trace.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if key in r["Kernel_Name"]:
                    dur += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        if dur > 0 and "sq" in rec and rec["sq"].get("GRBM_GUI_ACTIVE"):
            rec["clock_ghz_measured"] = rec["sq"]["GRBM_GUI_ACTIVE"] / 8.0 / dur
    except Exception as e:
        print("no clock estimate:", e)
    try:        # the bench line of the FETCH_SIZE pass: ADMM problem-iterations of the profiled step
        line = [l for l in open(os.path.join(root, "pmc_FETCH_SIZE.json")) if l.startswith("{")][-1]
        bj = json.loads(line)
        its = bj["aux"]["admm_iters_per_s"] * bj["ms_per_step"] * 1e-3 * bj["steps"]
        rec["problem_iterations"] = its
        rec["hbm_bytes_per_problem_iteration"] = rec["hbm_bytes_per_step"] / its
    except Exception as e:
        print("no bench line for the iteration count:", e)
    json.dump(rec, open(os.path.join(root, fname), "w"), indent=1)
    print("wrote", os.path.join(root, fname))

# ---- r04: the 7x20 step runs on two ADMM kernels (wavefront tier + row-local tail): one record with both
tiers = {}
for name, key in (("wavefront", "qp_admm_wv_kernel"), ("row_local", "qp_admm_rl_kernel")):
    ks = [k for k in out if key in k and "FETCH_SIZE" in out[k] and "WRITE_SIZE" in out[k]]
    if not ks:
        continue
    fetch = sum(out[k]["FETCH_SIZE"]["sum"] for k in ks); write = sum(out[k]["WRITE_SIZE"]["sum"] for k in ks)
    launches = sum(out[k]["FETCH_SIZE"]["dispatches"] for k in ks)
    tiers[name] = {"kernels": ks, "launches": launches, "fetch_KiB_raw": fetch, "write_KiB_raw": write,
                   "hbm_bytes": (2 * fetch + write) * 1024.0, "hbm_bytes_per_launch": (2 * fetch + write) * 1024.0 / max(launches, 1)}
    sq = {c: sum(out[k][c]["sum"] for k in ks if c in out[k]) for c in
          ("SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_ACTIVE_INST_VALU",
           "SQ_ACTIVE_INST_LDS", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE",
           "GRBM_GUI_ACTIVE") if any(c in out[k] for k in ks)}
    if sq:
        tiers[name]["sq"] = sq
    try:
        dur = 0
        for f in glob.glob(os.path.join(root, "pmc_sq2", "**", "*kernel_trace.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if key in r["Kernel_Name"]:
                    dur += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        if dur > 0 and sq.get("GRBM_GUI_ACTIVE"):
            tiers[name]["clock_ghz_measured"] = sq["GRBM_GUI_ACTIVE"] / 8.0 / dur
    except Exception as e:
        print("no clock estimate:", e)
if "wavefront" in tiers:
    rec = {"kernel_src_sha": bench.kernel_src_sha(), "batch": int(os.environ.get("PMC_BATCH", "1024")),
           "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (+ SQ passes), separate passes, bench.py --steps 1 --warmup 0 (scripts/gpu_pmc.sh)",
           "correction": "FETCH_SIZE x2 (gfx950 counts 64 B per 128-B request), WRITE_SIZE x1", "tiers": tiers,
           "hbm_bytes_per_launch": {k: v["hbm_bytes_per_launch"] for k, v in tiers.items()},
           "clock_ghz_measured": tiers["wavefront"].get("clock_ghz_measured")}
    try:
        line = [l for l in open(os.path.join(root, "pmc_FETCH_SIZE.json")) if l.startswith("{")][-1]
        bj = json.loads(line)
        t = bj["roofline"]["tiers"]
        for k in tiers:
            key2 = "wavefront" if k == "wavefront" else "row_local"
            its = t[key2]["problem_iterations_per_step"] * bj["steps"]
            tiers[k]["problem_iterations"] = its
            tiers[k]["hbm_bytes_per_problem_iteration"] = tiers[k]["hbm_bytes"] / max(its, 1)
    except Exception as e:
        print("no bench line for the per-tier iteration counts:", e)
    json.dump(rec, open(os.path.join(root, "traffic_r04.json"), "w"), indent=1)
    print("wrote", os.path.join(root, "traffic_r04.json"))
