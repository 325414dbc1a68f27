"""Developer runner (needs a GPU): one row of bench.py's graph-index table.  python scripts/run_graph_index.py [kind] [rows] [callers 0/1]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "meme-search-engine_amd")]
import torch  # noqa: E402,F401
import bench_ann as ba  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "hard"
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10_000_000
callers = bool(int(sys.argv[3])) if len(sys.argv) > 3 else False
print(json.dumps(ba.graph_index_bench(ROOT, kind, n, callers=callers)), flush=True)
