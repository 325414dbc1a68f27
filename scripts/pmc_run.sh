#!/bin/bash
# PMC passes over the bench (1e7 rows, few steps). Usage: pmc_run.sh <outdir-name> "<counters pass1>" "<counters pass2>" ...
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/$1; shift
mkdir -p $OUT
i=0
for c in "$@"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pass$i -o pmc -- python /root/repo/bench.py --rows 1e7 --steps 4 --warmup 1 --no-cpu-baseline --no-siglip > $OUT/pass$i.log 2>&1
  f=$(find $OUT/pass$i -name "*counter_collection.csv" | head -1)
  echo "== pass $i: $c -> $f"
  python - "$f" <<'PY'
import csv, sys, collections
f=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name'][:60]
    agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items():
    if 'scan_mfma' in k or 'scan_exact' in k:
        print(k, {c: (sum(x)/len(x)) for c,x in v.items()}, 'n=', len(next(iter(v.values()))))
PY
done
