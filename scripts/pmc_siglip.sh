#!/bin/bash
# PMC passes over the SigLIP bench (depth 2, batch 256). Usage: pmc_siglip.sh "<counters pass1>" ...
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/pmc_siglip
rm -rf $OUT; mkdir -p $OUT
i=0
for c in "$@"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pass$i -o pmc -- python /root/repo/scripts/siglip_bench.py 256 1 2 > $OUT/pass$i.log 2>&1
  f=$(find $OUT/pass$i -name "*counter_collection.csv" | head -1)
  echo "== pass $i: $c"
  python - "$f" <<'PY'
import csv, sys, collections
f=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name']
    k = k[k.find('gemm_kernel'):k.find('gemm_kernel')+14] if 'gemm_kernel' in k else ('attention' if 'attention_kernel' in k and 'pool' not in k else None)
    if k: agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(agg.items()):
    print(k, {c: round(sum(x)/len(x)) for c,x in v.items()}, 'n=', len(next(iter(v.values()))))
PY
done
