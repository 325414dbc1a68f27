#!/bin/bash
# the sustained-pattern half of scripts/profile_r05.sh alone -> gpurun_out/r05/pq_sustained_trace.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05
mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/pqs -o pq -- python $R/scripts/pq_trace_r05.py sustained > $OUT/pq_sustained.log 2>&1
python - $OUT/pqs <<'PY' > $OUT/pq_sustained_trace.txt
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "pq_scan64x4_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
calls = [rows[i:i + 8] for i in range(0, len(rows) - 7, 8)][2:]      # eight scans per 64-query call; the first two calls are warm-up
tot = sum((max(int(x["End_Timestamp"]) for x in c) - int(c[0]["Start_Timestamp"])) for c in calls)
n = sum(len(c) for c in calls)
print("pq_scan64x4_kernel<16, 8>: %d calls of eight scans; first scan's start to last scan's end / 8 = %.1f us per scan (sustained, two streams)" % (len(calls), tot / n / 1e3))
d = sorted(int(x["End_Timestamp"]) - int(x["Start_Timestamp"]) for c in calls for x in c)
print("the same launches' own trace durations (a launch's interval includes its wait for the other stream's scan): median %.1f us, min %.1f us" % (d[len(d) // 2] / 1e3, d[0] / 1e3))
PY
grep "HIP events" $OUT/pq_sustained.log >> $OUT/pq_sustained_trace.txt
cat $OUT/pq_sustained_trace.txt
rm -rf $OUT/pqs
