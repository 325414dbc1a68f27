#!/bin/bash
# rocprofv3 --kernel-trace of a few 64-query request-path calls (row-table entries, then shard-centroid entries): what a coalesced
# submission consists of on the device.  Needs a GPU: gpurun -- bash scripts/trace_request_path.sh
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/trace_rp
rocprofv3 --kernel-trace -d $R/gpurun_out/trace_rp -o rp --output-format csv -- python $R/scripts/request_path_small_calls.py 2e6 -64 > $R/gpurun_out/trace_rp.log 2>&1
tail -3 $R/gpurun_out/trace_rp.log
python - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/trace_rp/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def show(rs):
    t0 = int(rs[0]["Start_Timestamp"])
    for r in rs:
        print("%9.1f %8.1f  q=%s %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"][:90]))
# the last call of each entry rule ends with a select kernel after a beam_search kernel
idx = [i for i, r in enumerate(rows) if "beam_search_kernel" in r["Kernel_Name"]]
for which, name in ((len(idx) // 2 - 1, "row-table entries"), (len(idx) - 1, "shard-centroid entries")):
    i = idx[which]
    j = idx[which - 1] + 2 if which > 0 else 0
    print("# one 64-query call,", name, "(start us, duration us)")
    show(rows[j:i + 2])
PY
