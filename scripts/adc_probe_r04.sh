# developer probe: pq_adc_kernel beside a running scan, timing variants of the developer library (results WRONG for variants > 0)
#   0 shipped, 1 table copy only, 2 neighbouring code rows instead of gathered ones, 3 ids + code rows only (no table, no sums)
export MSE_HIP_LIB=$GRAFT_REPO_ROOT/meme-search-engine_amd/lib/libmse_hip_dev.so
for v in 1 2 3; do
  echo "== MSE_PQ_ADC_VARIANT=$v"
  MSE_PQ_ADC_VARIANT=$v bash $GRAFT_REPO_ROOT/scripts/trace_pq_r04.sh 2>&1 | grep -E "ms per call|pq_adc|pq_scan64x4" | tail -12
done
