"""The reference's recipe for a large index, on one GPU: k-means shard centroids (kmeans.py), every record spilled to its two
closest shards (dump_processor.rs:438-455), one Vamana graph per shard (generate-index-shard), lists merged per record
(dump_processor.rs:264-293), queries entered at the medioid of their closest shard (query_disk_index.rs:447-450) and searched
over the merged graph with the GPU-resident beam search.  Synthetic hierarchical rows as in graph_scale_bench.py.
usage: sharded_scale_bench.py [n_rows] [n_shards]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "meme-search-engine_amd"))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch  # noqa: E402
import mse  # noqa: E402
from mse import ffi  # noqa: E402
from graph_scale_bench import clustered, D  # noqa: E402

NONE = 0xFFFFFFFF


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    nq, K, R, L, batch = 1024, 10, 64, 192, 2048
    ffi.check(ffi.lib().mse_set_device(0))
    g0 = torch.Generator(device="cuda").manual_seed(0)
    hier = max(8, n // 5000)
    sup = torch.randn(hier, D, device="cuda", generator=g0)
    sup /= sup.norm(dim=1, keepdim=True)
    nc_ = max(64, n // 50)
    centres = sup[torch.randint(0, hier, (nc_,), device="cuda", generator=g0)] + torch.randn(nc_, D, device="cuda", generator=g0) * (0.7 / D ** 0.5)
    centres /= centres.norm(dim=1, keepdim=True)
    rows = clustered(n, centres, 0.3, 1)
    queries = clustered(nq, centres, 0.3, 2)
    del centres
    # shard centroids: a few k-means steps on a sample (max inner product assignment)
    samp = rows[torch.randperm(n, device="cuda", generator=g0)[:200_000]].float()
    cent = samp[:S].clone()
    for _ in range(8):
        a = (samp @ cent.T).argmax(dim=1)
        cent = torch.stack([samp[a == s].mean(dim=0) if (a == s).any() else cent[s] for s in range(S)])
    two = torch.empty(n, 2, dtype=torch.int64, device="cuda")
    for i in range(0, n, 1 << 20):
        two[i:i + (1 << 20)] = (rows[i:i + (1 << 20)].float() @ cent.T).topk(2, dim=1).indices
    merged = np.full((n, 2 * R), NONE, np.uint32)      # host: first R columns = the record's first shard, next R = its second
    seen = torch.zeros(n, dtype=torch.int8, device="cuda")
    specs = []
    t_build = 0.0
    for s in range(S):
        ids = torch.nonzero((two == s).any(dim=1)).squeeze(1)
        sub = rows[ids].contiguous()
        m = len(ids)
        vl = mse.VectorList.wrap_device(sub.data_ptr(), m, D, keepalive=sub)
        sr = mse.Searcher(vl)
        med = mse.medioid(vl)
        g = mse.BuildGraph(m, R)
        g.random_fill(s + 1)
        t0 = time.time()
        g.build(sr, np.random.default_rng(s).permutation(m).astype(np.uint32), med, mse.IndexBuildConfig(r=R, l=L, maxc=750), batch)
        t_build += time.time() - t0
        h = g.to_host()
        g.close()
        adj = torch.from_numpy(h.adj.astype(np.int64)).cuda()
        deg = torch.from_numpy(h.deg.astype(np.int64)).cuda()
        glob = ids[adj]                                                        # within-shard ids -> original ids
        glob[torch.arange(R, device="cuda")[None, :] >= deg[:, None]] = NONE
        col = seen[ids]                                                        # 0: first shard of the record, 1: second
        ids_h, glob_h, col_h = ids.cpu().numpy(), glob.cpu().numpy().astype(np.uint32), col.cpu().numpy()
        for c in (0, 1):
            pick = col_h == c
            merged[ids_h[pick], c * R:(c + 1) * R] = glob_h[pick]
        seen[ids] += 1
        specs.append(int(ids[med]))
        print(f"shard {s}: {m} records, built in {time.time()-t0:.1f} s", flush=True)
        del sub, vl, sr, adj, deg, glob
        torch.cuda.empty_cache()
    print(f"{S} shards, {t_build:.1f} s of graph building in total ({n/t_build:.0f} records/s, each record in two shards)", flush=True)
    # union per record, first occurrence kept (read_out_vertices): drop an entry equal to an earlier one of its row
    out = np.empty((n, 2 * R), np.uint32)
    degs = np.empty(n, np.uint32)
    for i in range(0, n, 1 << 16):
        blk = torch.from_numpy(merged[i:i + (1 << 16)].astype(np.int64)).cuda()
        dup = (blk[:, :, None] == blk[:, None, :]) & (torch.arange(2 * R, device="cuda")[None, :, None] > torch.arange(2 * R, device="cuda")[None, None, :])
        keep = (blk != NONE) & ~dup.any(dim=2)
        order = torch.argsort((~keep).to(torch.int8), dim=1, stable=True)      # kept entries first, original order
        blk = torch.gather(blk, 1, order)
        out[i:i + (1 << 16)] = blk.cpu().numpy().astype(np.uint32)
        degs[i:i + (1 << 16)] = keep.sum(dim=1).cpu().numpy()
    del merged
    torch.cuda.empty_cache()
    free, total = torch.cuda.mem_get_info()
    print(f"device memory free {free/1e9:.1f} GB before the merged graph ({out.nbytes/1e9:.1f} GB) is uploaded", flush=True)
    if out.nbytes + (6 << 30) > free:
        print("not enough device memory for the merged graph: stopping here", flush=True)
        return
    print(f"merged lists: mean {degs.mean():.1f} max {degs.max()} neighbours", flush=True)
    vecs = mse.VectorList.wrap_device(rows.data_ptr(), n, D, keepalive=rows)
    s = mse.Searcher(vecs)
    dgraph = mse.DeviceGraph(mse.IndexGraph(out, degs))
    qh = queries.cpu().numpy().view(np.uint16)
    _, truth = s.bruteforce_topk(qh, K)
    cents = cent.cpu().numpy().astype(np.float32)
    qf = queries.float().cpu().numpy()
    starts = np.array([specs[mse.select_shard(cents, qf[i])] for i in range(nq)], np.uint32)
    pq = codes = None      # neighbours are scored exactly: neither the codec nor the codes are touched
    for Ls in ((64, 100, 200, 400) if n >= 50_000_000 else (32, 64, 100, 200)):
        mse.disk_search_batch(s, pq, codes, dgraph, starts, qh, None, None, True, 4, Ls, 1024, as_arrays=True)
        t0 = time.perf_counter()
        res = mse.disk_search_batch(s, pq, codes, dgraph, starts, qh, None, None, True, 4, Ls, 1024, as_arrays=True)
        dt = time.perf_counter() - t0
        top = mse.topk_of_visited(res, K)
        hits = sum(len(set(top[i].tolist()) & set(truth[i].tolist())) for i in range(nq))
        print(f"L={Ls}: beam search over the merged graph (beam 4, exact neighbours, shard-selected entry) {nq/dt:8.0f} q/s recall@10 {hits/(K*nq):.3f} "
              f"({res['cmps'].mean():.0f} node fetches/query)", flush=True)


if __name__ == "__main__":
    main()
