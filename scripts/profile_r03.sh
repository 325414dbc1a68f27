#!/bin/bash
# Round-3 profile collection on the GPU box (run through gpurun): kernel-trace stats of the bench command, separate PMC passes
# (FETCH_SIZE | WRITE_SIZE, --kernel-trace only, as MI355X_MICROARCH.md's HBM section prescribes) for the scan at the headline
# configuration, kernel stats of the SigLIP forward on two streams AND on one stream, of the PQ scan (batched, pairs).
# Summaries land in gpurun_out/r03/ as text; copy them to profiles/.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03
mkdir -p $OUT
summarise() {   # <kernel_stats.csv> <header line> <out file>
python - "$1" "$2" "$3" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
with open(sys.argv[3], "w") as f:
    f.write(sys.argv[2] + "\n  calls    total_ms      avg_us       %  kernel\n")
    for r in rows[:24]:
        f.write("%7d %11.3f %11.3f %7.2f  %s\n" % (int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3,
                                                  float(r["Percentage"]), r["Name"][:150]))
PY
}
if [ "${1:-all}" = "all" ] || [ "$1" = "bench" ]; then
# 1. the bench command itself (all legs but the CPU baseline and the 1e7 graph leg, which add minutes of host work under the profiler)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -o b -- python $R/bench.py --no-cpu-baseline --no-graph-scale > $OUT/bench_profiled.json 2> $OUT/bench_profiled.err
summarise $(find $OUT/bench -name "*kernel_stats.csv" | head -1) "# rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-graph-scale   (MI355X, round 3: 1e8 x 1152 fp16, 256 queries/step, 20 steps + 3 warm-up, then 10 steps at 128 queries)" $OUT/r03_bench_1e8_kernel_stats.txt
python - $OUT/bench <<'PY' >> $OUT/r03_bench_1e8_kernel_stats.txt
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
d = {}
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "scan_mfma" in n:
        d.setdefault(n[:64], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for k, v in d.items():
    big = [x for x in v if x > 20]
    print("# %s: %d dispatches, %d over the 1e8-row index: avg %.3f ms (min %.3f, max %.3f)" % (k, len(v), len(big), sum(big) / max(len(big), 1), min(big or [0]), max(big or [0])))
PY
fi
if [ "${1:-all}" = "all" ] || [ "$1" = "traffic" ]; then
# 2. HBM traffic of the scan, one counter per pass
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -o p -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-siglip --no-pq --no-graph --no-graph-scale > $OUT/pmc_$c.log 2>&1
done
python - $OUT <<'PY' > $OUT/r03_pmc_traffic.json
import csv, glob, json, sys
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(sys.argv[1] + "/pmc_%s/**/*counter_collection.csv" % c, recursive=True)[0]
    per = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c and "scan_mfma" in r["Kernel_Name"]:
            per.setdefault(r["Kernel_Name"][:72], []).append(float(r["Counter_Value"]))
    out[c] = {k: [sum(v) / len(v), len(v)] for k, v in per.items()}
res = {"_comment": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) over `python bench.py --steps 4 --warmup 1` (scan legs only); averages per dispatch of the 256-query pass (scan_mfma2d_kernel) and of the 128-query pass (scan_mfma_kernel<3, 8>).  FETCH_SIZE is KiB and reports half of a wide coalesced stream on gfx950 (MI355X_MICROARCH.md, HBM section): bytes = FETCH_SIZE*1024*2 (calibrated in round 1 on scan_exact_kernel, profiles/r01_pmc_scan_1e7.txt).  WRITE_SIZE is uncalibrated (KiB*1024).",
       "raw": out, "rows": 100000000, "algorithmic_bytes_per_launch": 230400000000}
for c, key, mul in (("FETCH_SIZE", "hbm_read_bytes_per_launch", 2048.0), ("WRITE_SIZE", "hbm_write_bytes_per_launch", 1024.0)):
    for k, (v, n) in out[c].items():
        tag = "256" if "2d" in k else "128"
        res.setdefault("per_pass", {}).setdefault(tag, {"kernel": k})[key] = v * mul
hp = res.get("per_pass", {}).get("256", {})
res["queries_per_launch"] = 256
res["hbm_read_bytes_per_launch"] = hp.get("hbm_read_bytes_per_launch")
res["hbm_write_bytes_per_launch"] = hp.get("hbm_write_bytes_per_launch")
print(json.dumps(res, indent=1))
PY
fi
if [ "${1:-all}" = "all" ] || [ "$1" = "siglip" ]; then
# 3. SigLIP forward: two sub-batches on two streams (shipped), and one stream (kernel durations then add up to the wall time)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/siglip -o s -- python $R/scripts/siglip_bench.py 256 3 27 > $OUT/siglip.log 2>&1
summarise $(find $OUT/siglip -name "*kernel_stats.csv" | head -1) "# rocprofv3 --kernel-trace --stats -- python scripts/siglip_bench.py 256 3 27   (SigLIP SO400M/14-384 image tower, batch 256, bf16, seeded synthetic weights; 4 forwards, two sub-batches on two streams: kernel durations OVERLAP, their sum exceeds the wall time).  $(grep 'img/s' $OUT/siglip.log | tail -1)" $OUT/r03_siglip_b256_kernel_stats.txt
MSE_SIGLIP_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/siglip1 -o s -- python $R/scripts/siglip_bench.py 256 3 27 > $OUT/siglip1.log 2>&1
summarise $(find $OUT/siglip1 -name "*kernel_stats.csv" | head -1) "# MSE_SIGLIP_STREAMS=1 rocprofv3 --kernel-trace --stats -- python scripts/siglip_bench.py 256 3 27   (the same forward on ONE stream: kernels run back to back, the per-kernel durations are an attribution of the wall time; 4 forwards).  $(grep 'img/s' $OUT/siglip1.log | tail -1)" $OUT/r03_siglip_b256_one_stream_kernel_stats.txt
fi
if [ "${1:-all}" = "all" ] || [ "$1" = "pq" ]; then
# 4. PQ scan, batched (four queries share a pass), at BASELINE.md's 1e8 codes
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pq -o q -- python $R/scripts/pq_scan_bench.py 1e8 32 > $OUT/pq.log 2>&1
summarise $(find $OUT/pq -name "*kernel_stats.csv" | head -1) "# rocprofv3 --kernel-trace --stats -- python scripts/pq_scan_bench.py 1e8 32   (1e8 x 64-byte codes + 4 descriptor bytes, top-200; 11 one-query calls, then 4 calls of 32 queries = 32 four-query passes).  $(grep 'per query' $OUT/pq.log | tail -2 | tr '\n' ' ')" $OUT/r03_pq_scan_stats.txt
fi
ls -la $OUT/*.txt $OUT/*.json
