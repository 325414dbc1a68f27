# Where the GPU-resident beam search spends its time: kernel durations by launch size + SQ counters of `scripts/beam_batch_probe.py`.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_beam; rm -rf $OUT; mkdir -p $OUT
ROWS=${1:-2e6}
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o k -- python $R/scripts/beam_batch_probe.py $ROWS > $OUT/kt.log 2>&1
grep -E "queries per call|build s" $OUT/kt.log
python - $OUT/kt <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
agg = collections.defaultdict(list)
for r in rows:
    if "beam_search_kernel" in r["Kernel_Name"]:
        agg[(r["Kernel_Name"][:60], int(r.get("Grid_Size_X") or r["Grid_Size"]) // int(r.get("Workgroup_Size_X") or r["Workgroup_Size"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("# beam_search_kernel by launch: queries, launches, avg us, queries/s of the kernel alone")
for (k, q), v in sorted(agg.items(), key=lambda kv: kv[0][1]):
    print("%6d %4d %10.1f %12.0f  %s" % (q, len(v), sum(v) / len(v), q / (sum(v) / len(v)) * 1e6, k))
# everything else between the first and last beam kernel of the 2048-query calls
PY
if [ "$2" = "pmc" ]; then
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "FETCH_SIZE"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc_$tag -o p -- python $R/scripts/beam_batch_probe.py $ROWS > $OUT/pmc_$tag.log 2>&1
done
python - $OUT <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "beam_search_kernel" in r["Kernel_Name"]:
            agg[int(r.get("Grid_Size_X") or r["Grid_Size"]) // int(r.get("Workgroup_Size_X") or r["Workgroup_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for q, cs in sorted(agg.items()):
    print("beam_search_kernel,", q, "queries:")
    for c, v in sorted(cs.items()):
        print("   %-24s avg %16.0f  over %d dispatches" % (c, sum(v) / len(v), len(v)))
PY
fi
rm -rf $OUT/kt $OUT/pmc_*/
