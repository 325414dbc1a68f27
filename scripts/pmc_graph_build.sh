#!/bin/bash
# HBM traffic of the graph-build kernels: one rocprofv3 --pmc pass per counter over scripts/graph_build_bench.py.
# usage: pmc_graph_build.sh <outdir-name> <rows> [passes]
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/$1; ROWS=${2:-1e6}; PASSES=${3:-0}
mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -o pmc -- python /root/repo/scripts/graph_build_bench.py $ROWS 2048 $PASSES > $OUT/$c.log 2>&1
  f=$(find $OUT/$c -name "*counter_collection.csv" | head -1)
  echo "== $c -> $f"
  python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name']
    for key in ('graph_search_kernel<true', 'graph_search_kernelILb1', 'prune_kernel', 'backedge_gram', 'backedge_kernel'):
        if key in k:
            agg[key + ' ' + r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(agg.items()):
    print(f"{k}: launches {len(v)}, mean per launch {sum(v)/len(v):.4g}, total {sum(v):.4g}")
PY
done
grep -v amdgpu $OUT/FETCH_SIZE.log | grep "n=\|second"
