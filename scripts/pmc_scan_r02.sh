#!/bin/bash
# PMC passes (one counter group per run, --kernel-trace only) over the scan legs of bench.py at 1e7 rows:
# per-dispatch averages of the 256-query (NCT = 16) and 128-query (NCT = 8) scan kernels.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_scan_r02
mkdir -p $OUT
i=0
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_PENDING_STALL_CYCLES_sum" "SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pass$i -o pmc -- python $R/bench.py --rows 1e7 --steps 4 --warmup 1 --no-cpu-baseline --no-siglip --no-pq --no-graph --no-graph-scale > $OUT/pass$i.log 2>&1
done
python - $OUT <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "scan_mfma_kernel" in k:
            tag = "256q" if "<3, 16" in k else "128q" if "<3, 8" in k else k[:40]
            agg[r["Counter_Name"]][tag].append(float(r["Counter_Value"]))
print("# counter                          256-query pass        128-query pass     (averages per dispatch, 1e7 rows)")
for c in sorted(agg):
    v = agg[c]
    f = lambda t: ("%18.0f (n=%d)" % (sum(v[t]) / len(v[t]), len(v[t]))) if v.get(t) else "-"
    print("%-32s %s   %s" % (c, f("256q"), f("128q")))
PY
