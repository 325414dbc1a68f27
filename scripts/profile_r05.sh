#!/bin/bash
# Round-5 profiles -> gpurun_out/r05/ (scripts/collect_profiles_r05.py copies the summaries into profiles/):
#   bench_kernel_stats.txt       rocprofv3 --kernel-trace --stats of the bench command's scan legs (1e8 rows)
#   pq_burst_stats.txt           PQ scan, 42 eight-query calls on one stream: kernel stats
#   pq_sustained_trace.txt       PQ scan, 18 calls of 64 queries (two streams): per-call span from the trace's own timestamps
#   pmc_traffic.json             FETCH_SIZE / WRITE_SIZE passes (scan legs, PQ scan, SigLIP forward)
#   request_path_trace.txt       one 64-query request-path call per entry rule, kernel by kernel
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05
rm -rf $OUT; mkdir -p $OUT
stats() {  # $1 = dir with a *kernel_stats.csv, $2 = output text
python - $1 <<'PY' > $2
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print("  calls    total_ms      avg_us       %  kernel")
for r in rows[:22]:
    print("%7s %11.3f %11.3f %7s  %s" % (r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"], r["Name"][:160]))
PY
}
SCAN="--steps 12 --warmup 2 --no-siglip --no-pq --no-graph --no-graph-scale --no-cpu-baseline --no-callers --no-shard-point --no-ann-scale"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -o b -- python $R/bench.py $SCAN > $OUT/bench_profiled.json 2> $OUT/bench_profiled.err
stats $OUT/bench $OUT/bench_kernel_stats.txt; rm -rf $OUT/bench
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pqb -o pq -- python $R/scripts/pq_trace_r05.py burst > $OUT/pq_burst.log 2>&1
stats $OUT/pqb $OUT/pq_burst_stats.txt; rm -rf $OUT/pqb
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/pqs -o pq -- python $R/scripts/pq_trace_r05.py sustained > $OUT/pq_sustained.log 2>&1
python - $OUT/pqs <<'PY' > $OUT/pq_sustained_trace.txt
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "pq_scan64x4_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a call of 64 queries = eight scans, and a call ends before the next begins: consecutive groups of eight in start order
calls = [rows[i:i + 8] for i in range(0, len(rows) - 7, 8)][2:]      # the first two calls are warm-up
tot = sum((max(int(x["End_Timestamp"]) for x in c) - int(c[0]["Start_Timestamp"])) for c in calls)
n = sum(len(c) for c in calls)
print("pq_scan64x4_kernel<16, 8>: %d calls of eight scans; first scan's start to last scan's end / 8 = %.1f us per scan (sustained, two streams)" % (len(calls), tot / n / 1e3))
d = sorted(int(x["End_Timestamp"]) - int(x["Start_Timestamp"]) for c in calls for x in c)
print("the same launches' own trace durations (a launch's interval includes its wait for the other stream's scan): median %.1f us, min %.1f us" % (d[len(d) // 2] / 1e3, d[0] / 1e3))
PY
rm -rf $OUT/pqs
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -o pmc -- python $R/scripts/pq_trace_r05.py burst > $OUT/pmc_$c.log 2>&1
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmcs_$c -o pmc -- python $R/bench.py --steps 4 --warmup 1 ${SCAN#--steps 12 --warmup 2} > $OUT/pmcs_$c.log 2>&1
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmcg_$c -o pmc -- python $R/scripts/siglip_bench.py 256 2 27 > $OUT/pmcg_$c.log 2>&1
done
python $R/scripts/pmc_traffic_r05.py $OUT > $OUT/pmc_traffic.json
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmcs_FETCH_SIZE $OUT/pmcs_WRITE_SIZE $OUT/pmcg_FETCH_SIZE $OUT/pmcg_WRITE_SIZE
bash $R/scripts/trace_request_path.sh > $OUT/request_path_trace.txt 2>&1
ls -la $OUT
