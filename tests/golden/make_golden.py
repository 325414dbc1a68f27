"""Regenerates the committed fixtures under tests/golden/ from the CPU oracle.

The reference repository holds NO golden vectors, known-answer tests or fixtures for the
scoring path, and its Rust cannot be built in this image, so these
fixtures are outputs of OUR oracle (oracle/mse_oracle.c), which is itself pinned by the
hand-derived known answers in tests/test_oracle.py.  They exist so that (a) the oracle cannot
drift silently and (b) the HIP path is compared against committed numbers on the GPU box, where
/root/reference does not exist.

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import orc  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def bruteforce():
    base = orc.gen_rows_f16(0x5EED0001, 0, 256)
    queries = orc.gen_rows_f16(0x5EED0002, 0, 8)
    scores = np.stack([orc.score_all(base, q) for q in queries])
    top_s, top_i = orc.bruteforce_topk(base, queries, 10)
    # only the first 8 rows of the base are stored verbatim; the rest is regenerated from the seed
    np.savez_compressed(os.path.join(OUT, "bruteforce_256x1152.npz"), base_head=base[:8], queries=queries,
                        scores=scores, top_scores=top_s, top_ids=top_i, seed_base=0x5EED0001, seed_query=0x5EED0002)


def pq():
    rng = np.random.default_rng(11)
    d, dpc, nc = 1152, 18, 256
    lut = rng.standard_normal((d // dpc, nc)).astype(np.float32) * np.float32(0.03)
    codes = rng.integers(0, 256, size=(4096, d // dpc), dtype=np.uint8)
    desc = rng.integers(0, 256, size=(4096, 4), dtype=np.uint8)
    scales = (np.array([1.0, -0.5, 0.25, 2.0], np.float32) / np.float32(512.0))
    q = orc.PQ(np.zeros((nc, d), np.float32), np.eye(d, dtype=np.float32), dpc, d)
    adc = q.asymmetric_dot_product(lut, codes)
    adc_desc = q.adc_desc(lut, codes, desc, scales)
    np.savez_compressed(os.path.join(OUT, "pq_adc_4096.npz"), lut=lut, codes=codes, desc=desc, scales=scales, adc=adc,
                        adc_desc=adc_desc)


def neighbour_buffer():
    rng = np.random.default_rng(5)
    ops = []
    nb = orc.NeighbourBuffer(16)
    states = []
    for step in range(400):
        if rng.random() < 0.8:
            idx = int(rng.integers(0, 60))
            score = int(rng.integers(-50, 50)) * 1000  # many equal scores on purpose
            nb.insert(idx, score)
            ops.append((0, idx, score))
        else:
            r = nb.next_unvisited()
            ops.append((1, -1 if r is None else r, 0))
        states.append((nb.ids.copy(), nb.scores.copy(), nb.visited.copy()))
    maxlen = 16
    ids = np.full((len(states), maxlen), 0xFFFFFFFF, np.uint32)
    scores = np.zeros((len(states), maxlen), np.int64)
    visited = np.zeros((len(states), maxlen), np.uint8)
    lens = np.zeros(len(states), np.int32)
    for i, (a, b, c) in enumerate(states):
        lens[i] = len(a)
        ids[i, :len(a)], scores[i, :len(a)], visited[i, :len(a)] = a, b, c
    np.savez_compressed(os.path.join(OUT, "neighbour_buffer_trace.npz"), ops=np.array(ops, np.int64), ids=ids,
                        scores=scores, visited=visited, lens=lens, cap=16)


if __name__ == "__main__":
    orc.build()
    bruteforce()
    pq()
    neighbour_buffer()
    print("wrote", sorted(f for f in os.listdir(OUT) if f.endswith(".npz")))
