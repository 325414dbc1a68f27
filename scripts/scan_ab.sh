#!/bin/bash
# A/B of scan kernel variants on the GPU box: prints scan GB/s per variant
cd /root/repo
for v in "$@"; do
  env $v python bench.py --rows 1e7 --steps 10 --no-cpu-baseline --no-siglip 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['roofline']['achieved']), 'GB/s', round(d['value']), 'qps')"
done
