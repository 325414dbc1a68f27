#!/bin/bash
# A/B of the 256-query scan variants in the developer library (make dev): scan kernel time + answers vs the exact-order kernel
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export MSE_HIP_LIB=$R/meme-search-engine_amd/lib/libmse_hip_dev.so
ROWS=${ROWS:-1e7}
for v in ${VARIANTS:-"MSE_SCAN_2D=0" "MSE_SCAN_2D=16" "MSE_SCAN_2D=161"}; do
  env $v CHECK=1 python scripts/scan_ablate.py $ROWS 256 2>&1 | tail -3
done
