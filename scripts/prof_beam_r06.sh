# Round 6: the graph search's roofline on the HARD set at its operating point (L = 200, beam 4, 4096 queries per call, 1e7 rows):
# kernel durations (rocprofv3 --kernel-trace) and HBM bytes (--pmc FETCH_SIZE, a pass of its own) of beam_search_kernel, beside the
# searcher's own count of what the searches gathered (scripts/beam_hard_probe.py).  bash scripts/prof_beam_r06.sh [rows] [L] [beam]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_beam_hard; rm -rf $OUT; mkdir -p $OUT
ROWS=${1:-1e7}; L=${2:-200}; BEAM=${3:-4}
python $R/scripts/beam_hard_probe.py $ROWS $L $BEAM 4 > $OUT/plain.log 2>&1          # builds and caches the graph
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o k -- python $R/scripts/beam_hard_probe.py $ROWS $L $BEAM 4 > $OUT/kt.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o p -- python $R/scripts/beam_hard_probe.py $ROWS $L $BEAM 2 > $OUT/pmc_fetch.log 2>&1
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD --output-format csv -d $OUT/pmc_sq -o p -- python $R/scripts/beam_hard_probe.py $ROWS $L $BEAM 2 > $OUT/pmc_sq.log 2>&1
{
echo "# bash scripts/prof_beam_r06.sh $ROWS $L $BEAM on one MI355X (round 6): exactly scored request path, HARD set, one-pass Vamana graph (R 64, L 192)"
echo "# --- the probe alone (no profiler) ---"; cat $OUT/plain.log
echo "# --- under rocprofv3 --kernel-trace ---"; grep -v "^#" $OUT/kt.log
python - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(out + "/kt/**/*kernel_trace.csv", recursive=True)
if f:
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "beam_search_kernel" in r["Kernel_Name"]:
            wg = int(r.get("Workgroup_Size_X") or r["Workgroup_Size"])
            agg[(r["Kernel_Name"][:70], int(r.get("Grid_Size_X") or r["Grid_Size"]) // wg)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print("# beam_search_kernel by launch (kernel trace): queries, launches, avg us, queries/s of the kernel alone")
    for (k, q), v in sorted(agg.items(), key=lambda kv: kv[0][1]):
        print("%6d %4d %10.1f %12.0f  %s" % (q, len(v), sum(v) / len(v), q / (sum(v) / len(v)) * 1e6, k))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "beam_search_kernel" in r["Kernel_Name"]:
            wg = int(r.get("Workgroup_Size_X") or r["Workgroup_Size"])
            agg[int(r.get("Grid_Size_X") or r["Grid_Size"]) // wg][r["Counter_Name"]].append(float(r["Counter_Value"]))
for q, cs in sorted(agg.items()):
    print("# counters of beam_search_kernel,", q, "queries per launch (averages per dispatch):")
    for c, v in sorted(cs.items()):
        extra = ""
        if c == "FETCH_SIZE":
            extra = "   KiB x 2 (MI355X_MICROARCH.md: on gfx950 FETCH_SIZE tallies the 128-B requests of 16 B/lane loads at 64 B) = %.1f MB of HBM reads per launch" % (sum(v) / len(v) * 2048 / 1e6)
        print("   %-22s %16.0f  over %d dispatches%s" % (c, sum(v) / len(v), len(v), extra))
PY
} > $OUT/summary.txt 2>&1
rm -rf $OUT/kt $OUT/pmc_fetch $OUT/pmc_sq
cat $OUT/summary.txt
