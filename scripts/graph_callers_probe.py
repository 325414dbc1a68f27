"""Developer probe (needs a GPU): the request path's coalescer under T one-query request threads.
python scripts/graph_callers_probe.py [rows] [search_list]"""
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


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2_000_000
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    nq, K, R = 40960, 10, 64
    gen = ba.easy_generator(n)
    rows, queries = gen(n, 1), gen(nq, 2)
    torch.cuda.synchronize()
    vecs = mse.VectorList.wrap_device(rows.data_ptr(), n, ba.D, keepalive=rows)
    s = mse.Searcher(vecs)
    t0 = time.perf_counter()
    med = mse.medioid(vecs)
    g = mse.BuildGraph(n, R)
    g.random_fill(1)
    order = np.random.default_rng(3).permutation(n).astype(np.uint32)
    g.build(s, order, med, mse.IndexBuildConfig(r=R, l=192, maxc=750), 4096)
    print("build", time.perf_counter() - t0, flush=True)
    e_idx = np.sort(np.random.default_rng(5).choice(n, max(4096, n // 1500), replace=False)).astype(np.uint32)
    mse.set_entries(g, vecs, e_idx)
    qf = queries.float().cpu().numpy()
    qh = queries.cpu().numpy().view(np.uint16)
    _, truth = s.bruteforce_topk(qh, K)
    mse.disk_query_topk(s, None, None, g, qf[:4096], K, None, None, None, True, 4, L)
    t0 = time.perf_counter()
    mse.disk_query_topk(s, None, None, g, qf[:4096], K, None, None, None, True, 4, L)
    one_call = 4096 / (time.perf_counter() - t0)
    print("one call of 4096 f32 queries:", one_call, flush=True)
    for nb in (1, 16, 64, 256, 1024):
        mse.disk_query_topk(s, None, None, g, qf[:nb] if nb > 16 else qf[:nb], K, None, None, None, True, 4, L)
        t0 = time.perf_counter()
        for _ in range(20):
            mse.disk_query_topk(s, None, None, g, qf[:nb], K, None, None, None, True, 4, L)
        print("call of", nb, "queries: ms", (time.perf_counter() - t0) / 20 * 1e3, flush=True)
    def cg():
        out = {}
        for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu.stat"):
            try:
                out[f] = open(f).read().split()
            except Exception as e:  # noqa: BLE001
                out[f] = repr(e)
        return out
    print("nproc", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), cg(), flush=True)
    for nb in (1, 8, 16):
        ts = []
        for _ in range(40):
            t0 = time.perf_counter()
            mse.disk_query_topk(s, None, None, g, qf[:nb], K, None, None, None, True, 4, L)
            ts.append((time.perf_counter() - t0) * 1e3)
        print("call of", nb, "ms min/median/max", min(ts), sorted(ts)[20], max(ts), flush=True)
    # argv[3]: "affinity" = repeat the default setting with the process pinned to 16 / 32 / 64 cores (the box's CPU-time quota is 16)
    plan = [((0, 0, 0), None), ((2048, 200, 2), 32), ((4096, 200, 2), 32), ((4096, 200, 3), 32), ((0, 0, 0), 32)]
    if len(sys.argv) > 3 and sys.argv[3] == "quick":
        plan = [((0, 0, 0), 32), ((0, 0, 0), 32), ((0, 0, 0), 32)]
    if len(sys.argv) > 3 and sys.argv[3] == "workers":
        plan = [((0, 0, 3), 32), ((0, 0, 2), 32), ((0, 0, 3), 32), ((0, 0, 2), 32), ((0, 0, 4), 32)]
    if len(sys.argv) > 3 and sys.argv[3] == "affinity":
        plan = [((0, 0, 0), None), ((0, 0, 0), 16), ((0, 0, 0), 32), ((0, 0, 0), 64), ((4096, 1000, 2), 32), ((4096, 1000, 2), None), ((4096, 3000, 2), 32)]
    all_cores = sorted(os.sched_getaffinity(0))
    for co, cores in plan:
        os.sched_setaffinity(0, all_cores[:cores] if cores else all_cores)
        print("cores", cores, flush=True)
        out = ba.graph_callers(ROOT, vecs, g, qf, truth, L, K, 4, (64, 512, 4096), one_call, coalescer=co, pin_to_quota=False)
        print("cgroup", cg(), flush=True)
        print("tickets", json.dumps([{k: p[k] for k in ("in_flight", "host_threads", "query_copied_at_submit", "queries_per_s", "latency_ms", "queries_per_submission", "ms_per_submission", "all_answers_equal_the_batch_call", "vs_one_call_of_4096")} for p in out["tickets"]["points"]]), flush=True)
        print(json.dumps({"coalescer": co, "points": [{k: p[k] for k in ("threads", "queries_per_s", "latency_ms", "queries_per_submission", "ms_per_submission", "worker_seconds_in_submissions", "seconds", "all_answers_equal_the_batch_call", "vs_one_call_of_4096")} for p in out["points"]],
                          "perf_test": {k: out["perf_test_py_shape"][k] for k in ("queries_per_s", "latency_ms", "queries_per_submission")}}), flush=True)


if __name__ == "__main__":
    main()
