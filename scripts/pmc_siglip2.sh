#!/bin/bash
# PMC passes over the SigLIP bench (depth 2, batch 256), per-kernel averages.  Usage: pmc_siglip2.sh "<counters pass1>" ...
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/pmc_siglip2
rm -rf $OUT; mkdir -p $OUT
i=0
for c in "$@"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pass$i -o pmc -- python /root/repo/scripts/siglip_bench.py 256 1 2 > $OUT/pass$i.log 2>&1
  f=$(find $OUT/pass$i -name "*counter_collection.csv" | head -1)
  echo "== pass $i: $c"
  python - "$f" <<'PY'
import csv, sys, collections, re
f=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name']
    m=re.search(r'(gemm8pp_kernel<[^>]*>|attention64_kernel<[^>]*>|layernorm_kernel)', k)
    if not m:
        if 'layernorm_kernelIDF16' in k: name='layernorm_kernel<half>'
        else: continue
    else: name=m.group(1)
    agg[name][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(agg.items()):
    print(k, {c: round(sum(x)/len(x)) for c,x in v.items()}, 'n=', len(next(iter(v.values()))))
PY
done
