"""FETCH_SIZE / WRITE_SIZE passes of scripts/profile_r05.sh -> one JSON object (per-dispatch averages).  python scripts/pmc_traffic_r05.py <dir>"""
import csv, glob, json, sys, collections
def collect(prefix, match):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob(sys.argv[1] + "/" + prefix + c + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == c:
                    k = match(r["Kernel_Name"])
                    if k:
                        agg[k][c].append(float(r["Counter_Value"]))
    return agg
out = {"_comment": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only; scripts/profile_r05.sh): averages per dispatch.  FETCH_SIZE is KiB and reports half of a wide coalesced stream on gfx950 (MI355X_MICROARCH.md, HBM section): bytes = FETCH_SIZE * 1024 * 2 (calibrated in round 1 on scan_exact_kernel, profiles/r01_pmc_scan_1e7.txt); WRITE_SIZE is uncalibrated (KiB * 1024).  Scan legs: `python bench.py --steps 4 --warmup 1` (scan legs only, 1e8 rows); PQ: `python scripts/pq_trace_r05.py burst` (eight-query calls)."}
pq = collect("pmc_", lambda k: ("x8" if "x4_kernel<16, 8>" in k else "x4" if "x4" in k else "x1") if "pq_scan64" in k else None)
for k, name in (("x8", "pq_scan64x4"), ("x4", "pq_scan64x4_four_per_pass"), ("x1", "pq_scan64")):
    if pq[k]["FETCH_SIZE"]:
        f = pq[k]["FETCH_SIZE"]; w = pq[k]["WRITE_SIZE"] or [0.0]
        out[name] = {"vectors": 100000000, "algorithmic_bytes_per_launch": 6800000000, "dispatches": len(f),
                     "hbm_read_bytes_per_launch": sum(f) / len(f) * 2048, "hbm_write_bytes_per_launch": sum(w) / len(w) * 1024}
sc = collect("pmcs_", lambda k: "320" if "scan_mfma_kernel<2, 20" in k else "256" if "scan_mfma2d_kernel" in k else "128" if "scan_mfma_kernel<3, 8" in k else "192" if "scan_mfma_kernel<3, 12" in k else None)
out["rows"] = 100000000
out["algorithmic_bytes_per_launch"] = 230400000000
out["per_pass"] = {}
for k in ("320", "256", "192", "128"):
    if sc[k]["FETCH_SIZE"]:
        # only the full-size launches (the pick loop and the timed loop run at 1e8 rows)
        f = [v for v in sc[k]["FETCH_SIZE"] if v * 2048 > 1e11]; w = [v for v in sc[k]["WRITE_SIZE"] if v > 0] or [0.0]
        if f:
            out["per_pass"][k] = {"dispatches": len(f), "hbm_read_bytes_per_launch": sum(f) / len(f) * 2048, "hbm_write_bytes_per_launch": sum(w) / len(w) * 1024}
for k in ("320", "256"):   # the headline pass: the widest one measured
    if k in out["per_pass"]:
        out["queries_per_launch"] = int(k)
        out["hbm_read_bytes_per_launch"] = out["per_pass"][k]["hbm_read_bytes_per_launch"]
        out["hbm_write_bytes_per_launch"] = out["per_pass"][k]["hbm_write_bytes_per_launch"]
        break
# SigLIP image tower: every dispatch of the 3 forwards of `scripts/siglip_bench.py 256 2 27` (one warm-up + two timed), summed and divided by 3
sg = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    tot, names = 0.0, collections.Counter()
    for f in glob.glob(sys.argv[1] + "/pmcg_" + c + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c and "siglip" in r["Kernel_Name"]:
                tot += float(r["Counter_Value"]); names[r["Kernel_Name"][:60]] += 1
    sg[c] = tot / 3.0
if sg.get("FETCH_SIZE"):
    out["siglip"] = {"batch": 256, "depth": 27, "forwards_profiled": 3, "hbm_read_bytes_per_forward": sg["FETCH_SIZE"] * 2048,
                     "hbm_write_bytes_per_forward": sg["WRITE_SIZE"] * 1024,
                     "note": "all kernels of the mse::siglip namespace; FETCH_SIZE doubled as for the 16-byte-per-lane streams it was calibrated on (the GEMM and attention operand DMAs); WRITE_SIZE uncalibrated"}
print(json.dumps(out, indent=1))
