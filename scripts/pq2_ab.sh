#!/bin/bash
# A/B of the two-queries-per-pass PQ scan forms (developer library): batched per-query time at 2e7 codes
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export MSE_HIP_LIB=$R/meme-search-engine_amd/lib/libmse_hip_dev.so
for v in ${VARIANTS:-"X=0" "MSE_PQ2_DIRECT=12" "X=0" "MSE_PQ2_DIRECT=12"}; do
  env $v python bench.py --rows 1e6 --steps 2 --pq-rows ${PQ_ROWS:-2e7} --no-siglip --no-graph --no-graph-scale --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read())['pq_scan']; print('$v', 'ms/query batched', round(d['ms_per_query_batched'],4), 'single', round(d['ms_per_query'],4))"
done
