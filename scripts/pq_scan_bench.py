#!/usr/bin/env python3
"""PQ flat-scan timing alone (the `pq_scan` object of bench.py): python scripts/pq_scan_bench.py [rows] -> one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402  (before libmse_hip.so: one HIP runtime per process)
import bench  # noqa: E402


class A:
    pq_rows = float(sys.argv[1]) if len(sys.argv) > 1 else 1e8


print(json.dumps(bench.pq_bench(A)))
