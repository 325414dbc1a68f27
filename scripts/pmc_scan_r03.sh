#!/bin/bash
# PMC passes (one counter group per run, --kernel-trace only) over scripts/scan_ablate.py at 1e7 rows, developer library:
# per-dispatch averages per scan kernel variant.  Usage: pmc_scan_r03.sh "<env of variant 1>" "<env of variant 2>" ...
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_scan_r03
rm -rf $OUT; mkdir -p $OUT
export MSE_HIP_LIB=$R/meme-search-engine_amd/lib/libmse_hip_dev.so
GROUPS_=("GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_PENDING_STALL_CYCLES_sum" "SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS")
for v in "$@"; do
  tag=$(echo "$v" | tr ' =' '__')
  i=0
  for c in "${GROUPS_[@]}"; do
    i=$((i+1))
    env $v timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$tag/pass$i -o pmc -- python $R/scripts/scan_ablate.py 1e7 256 > $OUT/$tag.pass$i.log 2>&1
  done
done
python - $OUT "$@" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for v in sys.argv[2:]:
    tag = v.replace(" ", "_").replace("=", "_")
    agg = collections.defaultdict(list)
    for f in glob.glob(out + "/" + tag + "/pass*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "scan_mfma" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("# variant", v)
    for c in sorted(agg):
        print("%-32s %18.0f (n=%d)" % (c, sum(agg[c]) / len(agg[c]), len(agg[c])))
PY
