#!/bin/bash
# PMC passes (one counter group per run, --kernel-trace only) over scripts/pq_scan_bench.py: per-dispatch averages of the PQ scan kernels
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_pq_r04
rm -rf $OUT; mkdir -p $OUT
i=0
for c in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "FETCH_SIZE" "SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY" "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum SQ_INST_CYCLES_VMEM" ; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pass$i -o pmc -- python $R/scripts/pq_scan_bench.py 2e7 8 > $OUT/pass$i.log 2>&1
done
python - $OUT <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "pq_scan64" in k:
            tag = ("x8 (eight queries per pass, 8-bit tables, matrix-core sums)" if "x4_kernel<16, 8>" in k else
                   "x4 (four queries per pass, 12-bit tables, matrix-core sums)" if "x4" in k else
                   "x2 (two queries per pass)" if "x2" in k else "x1 (one query per pass)")
            agg[tag][r["Counter_Name"]].append(float(r["Counter_Value"]))
for tag in sorted(agg):
    print("#", tag)
    for c in sorted(agg[tag]):
        v = agg[tag][c]
        print("%-28s %16.0f (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
