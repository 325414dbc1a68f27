"""Developer runner (needs a GPU): bench.py's graph_index_1e8 leg alone.  python scripts/run_graph_1e8.py [rate_1e7] [budget_s] [rows] [easy|hard] [max passes]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "meme-search-engine_amd")]
import torch  # noqa: E402,F401
import bench_ann as ba  # noqa: E402

rate = float(sys.argv[1]) if len(sys.argv) > 1 else 2.8e5
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 1000.0
n = int(float(sys.argv[3])) if len(sys.argv) > 3 else 100_000_000
kind = sys.argv[4] if len(sys.argv) > 4 else "easy"
passes = int(sys.argv[5]) if len(sys.argv) > 5 else 2
print(json.dumps(ba.graph_index_1e8(ROOT, rate, budget, n=n, kind=kind, max_passes=passes)), flush=True)
