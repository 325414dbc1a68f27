# developer probe: the batched PQ scan without (parts of) its tail and on 1-3 streams (developer library; answers are garbage when a
# part of the tail is skipped) -- what the tail and the stream count cost the scans.  MSE_PQ_TAIL_SKIP bits: 1 tournament,
# 2 expand + re-score, 4 exact top-r, 8 certificate + re-score by rows
export MSE_HIP_LIB=$GRAFT_REPO_ROOT/meme-search-engine_amd/lib/libmse_hip_dev.so
for cfg in "15 1" "15 2" "15 3" "0 1" "0 3"; do
  set -- $cfg
  echo "== MSE_PQ_TAIL_SKIP=$1 MSE_PQ_LANES=$2"
  MSE_PQ_TAIL_SKIP=$1 MSE_PQ_LANES=$2 PQ_TRACE_NQ=64 PQ_TRACE_WINDOW_US=13000 bash $GRAFT_REPO_ROOT/scripts/trace_pq_r04.sh 2>&1 | grep -E "ms per call|pq_scan64x4" | cut -c1-60 | tail -9
done
