"""Developer probe (needs a GPU): hardness statistics and the search list a one-pass Vamana graph needs for recall@10 0.95 on the
synthetic sets of bench_ann.py, for a few parameter settings.  python scripts/hardness_probe.py [rows] [config ...]"""
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

CONFIGS = {
    "easy": None,
    "a": dict(cone=0.62, topic=0.45, within=0.55, noise=0.33, decay=0.6),
    "b": dict(cone=0.62, topic=0.35, within=0.65, noise=0.25, decay=0.6),
    "c": dict(cone=0.62, topic=0.45, within=0.55, noise=0.33, decay=0.3),
    "d": dict(cone=0.62, topic=0.30, within=0.70, noise=0.15, decay=0.8),
    "e": dict(cone=0.62, topic=0.25, within=0.75, noise=0.40, decay=0.5, rank=128),
    "f": dict(cone=0.50, topic=0.50, within=0.60, noise=0.50, decay=0.5, rank=64),
}


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2_000_000
    names = sys.argv[2:] or list(CONFIGS)
    nq, K, R = 2048, 10, 64
    for name in names:
        cfg = CONFIGS[name]
        t0 = time.perf_counter()
        if cfg is None:
            gen = ba.easy_generator(n)
            rows, queries = gen(n, 1), gen(nq, 2)
        else:
            hs = ba.HardSet(n, **cfg)
            rows, queries = hs.rows(n, 1), hs.rows(nq, 2)
        torch.cuda.synchronize()
        t_gen = time.perf_counter() - t0
        vecs = mse.VectorList.wrap_device(rows.data_ptr(), n, ba.D, keepalive=rows)
        s = mse.Searcher(vecs)
        stats, truth = ba.hardness(vecs, s, rows, queries)
        t0 = time.perf_counter()
        med = mse.medioid(vecs)
        g = mse.BuildGraph(n, R)
        g.random_fill(1)
        order = np.random.default_rng(3).permutation(n).astype(np.uint32)
        g.build(s, order, med, mse.IndexBuildConfig(r=R, l=192, maxc=750), 4096)
        t_build = time.perf_counter() - t0
        e_idx = np.sort(np.random.default_rng(5).choice(n, max(4096, n // 1500), replace=False)).astype(np.uint32)
        mse.set_entries(g, vecs, e_idx)
        qh = queries.cpu().numpy().view(np.uint16)
        sweep = []
        for L in (12, 16, 24, 32, 48, 64, 100, 150, 200, 300):
            mse.disk_query_topk(s, None, None, g, qh[:64], K, None, None, None, True, 4, L)
            t0 = time.perf_counter()
            top, _, st = mse.disk_query_topk(s, None, None, g, qh, K, None, None, None, True, 4, L)
            dt = time.perf_counter() - t0
            rec = ba.recall_at(top, truth)
            sweep.append((L, round(rec, 4), round(nq / dt), round(float(st["cmps"].mean()), 1)))
            if rec >= 0.99:
                break
        print(json.dumps({"config": name, "params": cfg, "rows": n, "gen_s": round(t_gen, 1), "build_s": round(t_build, 1), "stats": stats,
                          "sweep_L_recall_qps_fetches": sweep}), flush=True)
        g.close(); s.close(); vecs.close()
        del rows, queries
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
