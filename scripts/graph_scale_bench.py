"""Graph index at BASELINE config 3's size (1e7 x 1152): build the Vamana graph on the device, then queries/s and
recall@10 of the GPU-resident searches against the exact brute-force top-10 of the same index.
usage: graph_scale_bench.py [n_rows] [passes] [batch] [entries]
entries > 0: each query starts from the best of `entries` sampled rows (a stand-in for the reference's shard selection,
src/query_disk_index.rs:447-450: the medioid of the shard whose centroid is closest) instead of the one global medioid."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "meme-search-engine_amd"))
import torch  # noqa: F401,E402
import mse  # noqa: E402
from mse import ffi  # noqa: E402

D = 1152


def clustered(n, centres, noise, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    out = torch.empty(n, D, device="cuda", dtype=torch.float16)
    for i in range(0, n, 1 << 18):
        m = min(1 << 18, n - i)
        x = centres[torch.randint(0, len(centres), (m,), device="cuda", generator=g)] + torch.randn(m, D, device="cuda", generator=g) * (noise / D ** 0.5)
        out[i:i + m] = (x / x.norm(dim=1, keepdim=True)).half()
    return out


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
    passes = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
    n_entries = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    hier = int(sys.argv[5]) if len(sys.argv) > 5 else 0      # > 0: cluster centres are themselves drawn around `hier` super-centres
    nq, K, R, L = 1024, 10, 64, 192
    ffi.check(ffi.lib().mse_set_device(0))
    g0 = torch.Generator(device="cuda").manual_seed(0)
    nc_ = max(64, n // 50)
    if hier:
        sup = torch.randn(hier, D, device="cuda", generator=g0)
        sup /= sup.norm(dim=1, keepdim=True)
        centres = torch.empty(nc_, D, device="cuda")
        for i in range(0, nc_, 1 << 18):
            m = min(1 << 18, nc_ - i)
            centres[i:i + m] = sup[torch.randint(0, hier, (m,), device="cuda", generator=g0)] + torch.randn(m, D, device="cuda", generator=g0) * (0.7 / D ** 0.5)
        print(f"hierarchical centres: {nc_} centres around {hier} super-centres (noise 0.7)", flush=True)
    else:
        centres = torch.randn(nc_, D, device="cuda", generator=g0)
    centres /= centres.norm(dim=1, keepdim=True)
    rows = clustered(n, centres, 0.3, 1)
    queries = clustered(nq, centres, 0.3, 2)
    del centres
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free, total = torch.cuda.mem_get_info()
    print(f"rows resident; device memory free {free/1e9:.1f} of {total/1e9:.1f} GB", flush=True)
    need = n * R * 4 + n * 4 + batch * ((n + 31) // 32) * 4 + batch * 25000 * 12 + (2 << 30)
    if need > free:
        print(f"not enough device memory for the graph and the visited sets ({need/1e9:.1f} GB needed): stopping here", flush=True)
        return
    vecs = mse.VectorList.wrap_device(rows.data_ptr(), n, D, keepalive=rows)
    s = mse.Searcher(vecs)
    t0 = time.time()
    med = mse.medioid(vecs)
    print(f"n={n}: medioid {time.time()-t0:.1f} s", flush=True)
    g = mse.BuildGraph(n, R)
    g.random_fill(1)
    rng = np.random.default_rng(3)
    alphas = [int(x) for x in os.environ.get("ALPHAS", "65536").split(",")]   # relaxation factor (x 2^16) per pass

    def build_pass(p):
        order = rng.permutation(n).astype(np.uint32)
        t0 = time.time()
        seg = max(batch, (n // 20 + batch - 1) // batch * batch)      # progress lines; a multiple of the batch, so the result is the same
        for o0 in range(0, n, seg):
            g.build(s, order[o0:o0 + seg], med, mse.IndexBuildConfig(r=R, l=L, maxc=750, alpha=alphas[min(p, len(alphas) - 1)]), batch)
            if n >= 5_000_000:
                print(f"  {min(n, o0 + seg)} points in {time.time()-t0:.0f} s", flush=True)
        dt = time.time() - t0
        print(f"pass {p + 1}: {dt:.1f} s = {n/dt:.0f} points/s (R {R}, L {L}, C 750, batch {batch}, alpha {alphas[min(p, len(alphas) - 1)]})", flush=True)

    qh = queries.cpu().numpy().view(np.uint16)
    t0 = time.time()
    _, truth = s.bruteforce_topk(qh, K)
    print(f"brute-force truth for {nq} queries: {time.time()-t0:.2f} s", flush=True)
    disk = n <= 20_000_000 or os.environ.get("DISK_VARIANT") == "1"   # the disk variant wants a codes array (64 B per row) even when neighbours are scored exactly
    if disk:
        cents = (np.random.default_rng(4).standard_normal((256, D)) / np.sqrt(D)).astype(np.float32)
        pq = mse.ProductQuantizer(cents, np.eye(D, dtype=np.float32), 18, D)     # unused in exact-neighbour mode
        codes = mse.Codes(np.zeros((n, 64), np.uint8), None)
    med_starts = np.full(nq, med, np.uint32)
    starts = None
    if n_entries:
        eid = np.sort(np.random.default_rng(9).choice(n, n_entries, replace=False)).astype(np.int64)
        erows = rows[torch.from_numpy(eid).cuda()].contiguous()
        es = mse.Searcher(mse.VectorList.wrap_device(erows.data_ptr(), n_entries, D, keepalive=erows))
        t0 = time.perf_counter()
        _, best = es.bruteforce_topk(qh, 1)
        starts = eid[best[:, 0].astype(np.int64)].astype(np.uint32)
        print(f"entry points: best of {n_entries} sampled rows per query ({(time.perf_counter()-t0)*1e3:.1f} ms for {nq} queries)", flush=True)

    def sweep(starts, label):
        print(f"-- start: {label}", flush=True)
        for Ls in ((100, 200, 400, 800) if n >= 50_000_000 else (32, 64, 100, 200)):
            if not disk:
                g.search_batch(s, starts, qh, Ls, as_arrays=True)
                t0 = time.perf_counter()
                rid, _, _, nd = g.search_batch(s, starts, qh, Ls, as_arrays=True)
                dr = time.perf_counter() - t0
                rh = sum(len(set(rid[i, :K].tolist()) & set(truth[i].tolist())) for i in range(nq))
                print(f"L={Ls}: in-RAM greedy search {nq/dr:8.0f} q/s recall@10 {rh/(K*nq):.3f} ({nd.mean():.0f} distances/query)", flush=True)
                continue
            mse.disk_search_batch(s, pq, codes, g, starts, qh, None, None, True, 4, Ls, 1024, as_arrays=True)   # warm: scratch is allocated on first use
            t0 = time.perf_counter()
            res = mse.disk_search_batch(s, pq, codes, g, starts, qh, None, None, True, 4, Ls, 1024, as_arrays=True)
            dt = time.perf_counter() - t0
            top = mse.topk_of_visited(res, K)
            hits = sum(len(set(top[i].tolist()) & set(truth[i].tolist())) for i in range(nq))
            g.search_batch(s, starts, qh, Ls, as_arrays=True)
            t0 = time.perf_counter()
            rid, _, _, _ = g.search_batch(s, starts, qh, Ls, as_arrays=True)
            dr = time.perf_counter() - t0
            rh = sum(len(set(rid[i, :K].tolist()) & set(truth[i].tolist())) for i in range(nq))
            print(f"L={Ls}: beam search (beam 4, exact neighbours) {nq/dt:8.0f} q/s recall@10 {hits/(K*nq):.3f} "
                  f"({res['cmps'].mean():.0f} node fetches/query); in-RAM greedy search {nq/dr:8.0f} q/s recall@10 {rh/(K*nq):.3f}", flush=True)

    for p in range(passes):
        build_pass(p)
        sweep(med_starts, f"the medioid, after pass {p + 1}")
        if starts is not None:
            sweep(starts, f"best of {n_entries} sampled rows, after pass {p + 1}")


if __name__ == "__main__":
    main()
