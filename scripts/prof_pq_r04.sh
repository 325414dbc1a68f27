#!/bin/bash
# kernel stats of the PQ flat-scan bench (scripts/pq_scan_bench.py) -> gpurun_out/prof_pq_r04/
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_pq_r04
rm -rf $OUT; mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o pq -- python $R/scripts/pq_scan_bench.py ${1:-1e8} > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log > $OUT/bench_line.json
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python - $f <<'PY' | tee $OUT/kernel_stats.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("  calls    total_ms      avg_us       %  kernel")
for r in rows[:18]:
    print("%7s %11.3f %11.3f %7s  %s" % (r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"], r["Name"][:150]))
PY
