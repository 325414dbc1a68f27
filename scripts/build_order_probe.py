"""Developer probe (needs a GPU): does the ORDER of the points inside a build batch matter for the build rate?  The insertion order is a
free input (the reference shuffles, diskann/src/lib.rs:287-296); the batches stay the same random subsets, only the position of a
point inside its batch changes: sorted by coarse cluster, so that workgroups that run side by side search towards the same region.
python scripts/build_order_probe.py [kind] [rows]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "meme-search-engine_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
import mse  # noqa: E402
import bench_ann as ba  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "hard"
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 4_000_000
K, R, nq = 10, 64, 2048
if kind == "easy":
    gen = ba.easy_generator(n)
    rows, queries = gen(n, 1), gen(nq, 2)
else:
    hs = ba.HardSet(n, **ba.HARD_PARAMS)
    rows, queries = hs.rows(n, 1), hs.rows(nq, 2)
torch.cuda.synchronize()
vecs = mse.VectorList.wrap_device(rows.data_ptr(), n, ba.D, keepalive=rows)
s = mse.Searcher(vecs)
qh = queries.cpu().numpy().view(np.uint16)
_, truth = s.bruteforce_topk(qh, K)
med = mse.medioid(vecs)
t0 = time.perf_counter()
cen, _ = ba.shard_centroid_entries(rows, n, n_shards=256)
cen_t = torch.from_numpy(cen).cuda().half()
asg = torch.empty(n, dtype=torch.int32, device="cuda")
for i in range(0, n, 1 << 20):
    asg[i:i + (1 << 20)] = torch.argmax(rows[i:i + (1 << 20)] @ cen_t.T, dim=1).int()
asg = asg.cpu().numpy()
print("coarse assignment seconds", time.perf_counter() - t0, flush=True)
perm = np.random.default_rng(3).permutation(n).astype(np.uint32)
e_idx = np.sort(np.random.default_rng(5).choice(n, max(4096, n // 1500), replace=False)).astype(np.uint32)
for batch, sort_in_batch in ((4096, False), (4096, True), (16384, True), (16384, False), (65536, True)):
    order = perm.copy()
    if sort_in_batch:
        for b0 in range(0, n, batch):
            seg = order[b0:b0 + batch]
            order[b0:b0 + batch] = seg[np.argsort(asg[seg], kind="stable")]
    g = mse.BuildGraph(n, R)
    g.random_fill(1)
    t0 = time.perf_counter()
    g.build(s, order, med, mse.IndexBuildConfig(r=R, l=192, maxc=750), batch)
    dt = time.perf_counter() - t0
    mse.set_entries(g, vecs, e_idx)
    rec = {}
    for L in (32, 64, 100):
        top, _, _ = mse.disk_query_topk(s, None, None, g, qh, K, None, None, None, True, 4, L)
        rec[L] = round(ba.recall_at(top, truth), 4)
    print(json.dumps({"kind": kind, "rows": n, "batch": batch, "sorted_in_batch": sort_in_batch, "build_s": round(dt, 2), "points_per_s": round(n / dt),
                      "recall_by_L": rec}), flush=True)
    g.close()
