cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03pq
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pq -o q -- python $R/scripts/pq_scan_bench.py 2e7 32 > $OUT/pq.log 2>&1
python - $(find $OUT/pq -name "*kernel_stats.csv" | head -1) <<'PY' > $OUT/stats.txt
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:24]:
    print("%7d %11.3f %11.3f %7.2f  %s" % (int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, float(r["Percentage"]), r["Name"][:120]))
PY
cat $OUT/pq.log | tail -3 >> $OUT/stats.txt
