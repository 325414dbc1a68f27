#!/bin/bash
# rocprofv3 --kernel-trace of small-batch SigLIP forwards: where a batch-1 forward's time goes.  gpurun -- bash scripts/trace_siglip_latency.sh [batch]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
B=${1:-1}
rm -rf $R/gpurun_out/trace_sl
rocprofv3 --kernel-trace -d $R/gpurun_out/trace_sl -o sl --output-format csv -- python $R/scripts/siglip_latency_trace.py $B > $R/gpurun_out/trace_sl.log 2>&1
grep batch $R/gpurun_out/trace_sl.log
python - <<PY
import csv, glob, collections
f = glob.glob("$R/gpurun_out/trace_sl/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = n.replace("mse::siglip::", "").replace("(anonymous namespace)::", "")
    return n[:70]
# the last text forward = the last run of kernels ending with l2norm before the image engine's first kernels; find l2norm kernels
l2 = [i for i, r in enumerate(rows) if "l2norm" in r["Kernel_Name"]]
def forward(end_idx, start_after):
    return rows[start_after + 1:end_idx + 1]
text_fw = forward(l2[7], l2[6])      # 3 warm + 5 timed text forwards -> l2[0..7]
img_fw = forward(l2[-1], l2[-2])
for name, fw in (("text", text_fw), ("image", img_fw)):
    t0, t1 = int(fw[0]["Start_Timestamp"]), int(fw[-1]["End_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in fw)
    print("# %s tower, batch $B: %d kernels, first start to last end %.3f ms, kernels busy %.3f ms" % (name, len(fw), (t1 - t0) / 1e6, busy / 1e6))
    agg = collections.OrderedDict()
    for r in fw:
        k = short(r["Kernel_Name"])
        a = agg.setdefault(k, [0, 0])
        a[0] += 1; a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("   %4d x %9.1f us avg  %9.3f ms  %s" % (c, t / c / 1e3, t / 1e6, k))
    print("   first layer, kernel by kernel (start us, duration us):")
    for r in fw[:12]:
        print("      %9.1f %8.1f  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, short(r["Kernel_Name"])))
PY
