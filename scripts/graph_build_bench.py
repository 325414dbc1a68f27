"""Vamana build on the device: points/s for the passes of generate_index_shard (first pass alpha 1.0, optional second
pass alpha 1.2), then recall@1 of self-queries and recall@10 of outside queries on the built graph.
usage: graph_build_bench.py [n_rows] [batch] [second_pass 0/1] [L] [R]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "meme-search-engine_amd"))
import torch  # noqa: F401,E402
import mse  # noqa: E402
from mse import ffi  # noqa: E402


def clustered(n, d=1152, n_centres=4096, noise=0.3, seed=0):
    """SURVEY 8(d)'s clustered set: unit centres + N(0, noise^2/d) noise, renormalised (made with torch on the device)."""
    g = torch.Generator(device="cuda").manual_seed(seed)
    c = torch.randn(n_centres, d, device="cuda", generator=g)
    c /= c.norm(dim=1, keepdim=True)
    out = torch.empty(n, d, device="cuda", dtype=torch.float16)
    for i in range(0, n, 1 << 18):
        m = min(1 << 18, n - i)
        x = c[torch.randint(0, n_centres, (m,), device="cuda", generator=g)] + torch.randn(m, d, device="cuda", generator=g) * (noise / d ** 0.5)
        out[i:i + m] = (x / x.norm(dim=1, keepdim=True)).half()
    return out


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100000
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    second = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    L = int(sys.argv[4]) if len(sys.argv) > 4 else 192
    R = int(sys.argv[5]) if len(sys.argv) > 5 else 64
    ffi.check(ffi.lib().mse_set_device(0))
    rows = clustered(n)
    vecs = mse.VectorList.wrap_device(rows.data_ptr(), n, 1152, keepalive=rows)
    s = mse.Searcher(vecs)
    med = mse.medioid(vecs)
    g = mse.BuildGraph(n, R)
    t0 = time.time()
    g.random_fill(1)
    t_fill = time.time() - t0
    order = np.random.default_rng(0).permutation(n).astype(np.uint32)
    cfg = mse.IndexBuildConfig(r=R, l=L, maxc=750)
    # small warm-up batch (module load, buffers)
    t0 = time.time()
    g.build(s, order, med, cfg, batch)
    t1 = time.time() - t0
    print(f"n={n} L={L} R={R} batch={batch}: random fill {t_fill*1e3:.1f} ms; first pass {t1:.2f} s = {n/t1:.0f} points/s", flush=True)
    if second:
        cfg2 = mse.IndexBuildConfig(r=R, l=L, maxc=750, alpha=int(os.environ.get("ALPHA2", "65536")))
        t0 = time.time()
        g.build(s, order, med, cfg2, batch)
        t2 = time.time() - t0
        print(f"second pass (alpha_2 {cfg2.alpha}) {t2:.2f} s = {n/t2:.0f} points/s", flush=True)
    h = g.to_host()
    print(f"degree: mean {h.deg.mean():.1f} min {h.deg.min()} max {h.deg.max()}")
    nq = 1000
    qi = np.random.default_rng(1).choice(n, nq, replace=False)
    q = rows[torch.from_numpy(qi).cuda()].cpu().numpy().view(np.uint16)
    t0 = time.time()
    res = g.search_batch(s, med, q, L)
    ts = time.time() - t0
    r1 = np.mean([res[k][0][0] == qi[k] for k in range(nq)])
    print(f"self-query recall@1 {r1:.3f} (L={L}); {nq/ts:.0f} q/s; mean distances {np.mean([r[2] for r in res]):.0f}")
    # outside queries: noisy copies of base rows; truth by brute force
    qo = clustered(nq, seed=5).cpu().numpy().view(np.uint16)
    truth_s, truth_i = s.bruteforce_topk(qo, 10, mse.MODE_AUTO)
    res = g.search_batch(s, med, qo, L)
    rec = np.mean([len(set(res[k][0][:10].tolist()) & set(truth_i[k].tolist())) / 10 for k in range(nq)])
    print(f"outside-query recall@10 {rec:.3f} (L={L})")


if __name__ == "__main__":
    main()
