"""Synthetic datasets for the approximate-search legs of bench.py, their hardness statistics, and the graph index's request path in
the reference's call shape.  Imported by bench.py (and scripts/hardness_probe.py); everything runs on the device through torch
(synthetic rows) and the C ABI (searches).

Three sets, all unit-norm fp16 rows of width 1152:
  easy   the round-1..4 set: rows/50 centres around rows/5000 super-centres, noise 0.3 -- the true top-10 of a query are its ~50
         cluster-mates; a beam search reaches recall 0.98 with a search list of 12
  hard   no micro-clusters: a common mean direction (the "cone" of contrastive embeddings: random pairs have cosine ~ 0.45), a
         low-rank Gaussian with a power-law spectrum (rank 96) around power-law sized topic centres, isotropic noise on top.
         Neighbours are separated from non-neighbours by a margin of a few hundredths of cosine, not by 0.3
  ood    the hard base set queried from ANOTHER distribution, as text queries against image embeddings are
         (src/generate_index_shard.rs:62-84,127-131): queries = the hard mixture with its own topic mass, pushed along a fixed
         "modality gap" direction and with more isotropic noise; the graph is built with a query sample appended after the base rows
         (query_breakpoint) and robust_stitch (diskann/src/lib.rs:326-374), as the reference's OOD-DiskANN variant does
"""
import math
import time

D = 1152


def easy_generator(n):
    """rows/50 centres around rows/5000 super-centres, noise 0.3 -> f(m, seed) returning an [m, 1152] fp16 device tensor."""
    import torch
    g0 = torch.Generator(device="cuda").manual_seed(0)
    hier = max(8, n // 5000)
    sup = torch.randn(hier, D, device="cuda", generator=g0)
    sup /= sup.norm(dim=1, keepdim=True)
    nc_ = max(64, n // 50)
    centres = sup[torch.randint(0, hier, (nc_,), device="cuda", generator=g0)] + torch.randn(nc_, D, device="cuda", generator=g0) * (0.7 / D ** 0.5)
    centres /= centres.norm(dim=1, keepdim=True)

    def clustered(m, seed):
        g = torch.Generator(device="cuda").manual_seed(seed)
        out = torch.empty(m, D, device="cuda", dtype=torch.float16)
        for i in range(0, m, 1 << 18):
            c = min(1 << 18, m - i)
            x = centres[torch.randint(0, nc_, (c,), device="cuda", generator=g)] + torch.randn(c, D, device="cuda", generator=g) * (0.3 / D ** 0.5)
            out[i:i + c] = (x / x.norm(dim=1, keepdim=True)).half()
        return out

    return clustered


class HardSet:
    """normalise(cone * mu + topic centre + within-topic low-rank Gaussian + isotropic noise).  Parameters are fractions of the
    squared norm before normalisation: cone^2 + topic^2 + within^2 + noise^2 = 1."""

    def __init__(self, n, rank=96, n_topics=None, cone=0.62, topic=0.45, within=0.55, noise=0.33, decay=0.6, zipf=1.0, seed=0):
        import torch
        g0 = torch.Generator(device="cuda").manual_seed(1000 + seed)
        self.rank = rank
        self.n_topics = n_topics or max(64, int(round(n ** 0.5 / 2)))
        s = math.sqrt(cone ** 2 + topic ** 2 + within ** 2 + noise ** 2)
        self.cone, self.topic, self.within, self.noise = cone / s, topic / s, within / s, noise / s
        q, _ = torch.linalg.qr(torch.randn(D, rank + 2, device="cuda", generator=g0))
        self.mu = q[:, 0].contiguous()                       # the cone axis
        self.gap = q[:, 1].contiguous()                      # the modality-gap direction of the OOD queries (orthogonal to everything else)
        basis = q[:, 2:].contiguous()                        # [D, rank] orthonormal
        spec = torch.arange(1, rank + 1, device="cuda", dtype=torch.float32) ** (-decay)
        spec /= spec.norm()                                  # power-law spectrum, unit total energy
        self.A = (basis * spec[None, :]).contiguous()        # z ~ N(0, I_rank) -> A z has unit expected squared norm
        z = torch.randn(self.n_topics, rank, device="cuda", generator=g0)
        self.centres = z @ self.A.T                          # topic centres live in the low-rank subspace
        w = torch.arange(1, self.n_topics + 1, device="cuda", dtype=torch.float32) ** (-zipf)
        self.topic_p = w / w.sum()                           # power-law topic sizes
        wq = w[torch.randperm(self.n_topics, device="cuda", generator=g0)]
        self.topic_p_queries = wq / wq.sum()                 # OOD queries: another topic mass over the same topics

    def rows(self, m, seed, queries=None, gap=0.0, extra_noise=0.0):
        """[m, 1152] fp16 device tensor.  queries='ood': topic mass of the query distribution, pushed `gap` along the gap direction,
        `extra_noise` more isotropic noise."""
        import torch
        g = torch.Generator(device="cuda").manual_seed(seed)
        out = torch.empty(m, D, device="cuda", dtype=torch.float16)
        p = self.topic_p_queries if queries == "ood" else self.topic_p
        noise = math.sqrt(self.noise ** 2 + extra_noise ** 2)
        for i in range(0, m, 1 << 18):
            c = min(1 << 18, m - i)
            t = torch.multinomial(p, c, replacement=True, generator=g)
            z = torch.randn(c, self.rank, device="cuda", generator=g)
            x = self.cone * self.mu[None, :] + self.topic * self.centres[t] + self.within * (z @ self.A.T)
            x += torch.randn(c, D, device="cuda", generator=g) * (noise / D ** 0.5)
            if gap:
                x += gap * self.gap[None, :]
            out[i:i + c] = (x / x.norm(dim=1, keepdim=True)).half()
        return out


def hardness(vecs, searcher, rows, queries_f16, k=10, k_lid=20, n_random=4096):
    """Statistics of a (base, query) pair that say how hard approximate search is, from exact brute force on `queries_f16`:
      relative_contrast   mean dot of a random base row with the query / ... reported as both means and the classic ratio in DISTANCE
                          form: mean distance to a random row / mean distance to the k-th neighbour (distance = 1 - dot; unit rows)
      lid_mle             Levina-Bickel / Amsaleg MLE of the local intrinsic dimension at k_lid neighbours, on Euclidean distances
                          sqrt(2 - 2 dot), averaged over the queries
    Returns (stats dict, exact top-k ids)."""
    import numpy as np
    import torch
    import mse
    q = queries_f16
    nq = q.shape[0]
    sc, ids = searcher.bruteforce_topk(q.cpu().numpy().view(np.uint16), max(k, k_lid))
    dots = sc.astype(np.float64) / float(mse.SCALE)
    sel = torch.randint(0, rows.shape[0], (n_random,), device="cuda")
    rnd = (q.float() @ rows[sel].float().T).cpu().numpy().astype(np.float64)        # [nq, n_random]
    d_k = 1.0 - dots[:, k - 1]
    d_rand = 1.0 - rnd.mean(axis=1)
    r = np.sqrt(np.maximum(2.0 - 2.0 * dots[:, :k_lid], 1e-12))                     # ascending distances
    lid = -1.0 / np.mean(np.log(np.maximum(r[:, :-1], 1e-12) / r[:, -1:]), axis=1)
    return ({"queries": int(nq), "mean_dot_nn1": float(dots[:, 0].mean()), f"mean_dot_nn{k}": float(dots[:, k - 1].mean()),
             "mean_dot_random_row": float(rnd.mean()), "relative_contrast_at_%d" % k: float(d_rand.mean() / d_k.mean()),
             "lid_mle_k%d" % k_lid: float(np.median(lid)), "note": "relative contrast = mean (1 - dot) to a random row / mean (1 - dot) to the %d-th neighbour; "
             "LID = MLE over %d neighbours on Euclidean distances, median over queries" % (k, k_lid)}, ids[:, :k])


def recall_at(top, truth):
    k = truth.shape[1]
    return sum(len(set(top[i, :k].tolist()) & set(truth[i].tolist())) for i in range(truth.shape[0])) / (k * truth.shape[0])


def _callers_lib(root):
    import ctypes as C
    import os
    import subprocess
    so = os.path.join(root, "scripts", "native", "libmse_callers.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.dirname(so)], stdout=subprocess.DEVNULL)
    H = C.CDLL(so)
    H.mse_callers_run_async.restype = C.c_double
    H.mse_callers_run_async.argtypes = [C.c_void_p] * 10 + [C.c_size_t, C.c_size_t, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int,
                                                             C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
    H.mse_callers_run_query.restype = C.c_double
    H.mse_callers_run_query.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
    return H


def graph_callers(root, vecs, g, queries_f32, truth, search_list, k=10, beam=4, thread_counts=(64, 512, 4096), one_call_qps=None,
                  pq=None, codes=None, disable_pq=True, rounds=None, coalescer=(0, 0, 0), pin_to_quota=True):
    """The metric's path in the reference's call shape (src/query_disk_index.rs:436-540,711-736): T native request threads, closed
    loop, ONE f32 query per mse_disk_query_topk_f32 call through host pointers (entry step, f16 copy, greedy_search, top-k of the
    visited records -- all inside the call); the calls meet in the graph's coalescer.  Every answer is compared with the batch call's
    answer for the same query.  Plus the shape of the repo's own load test (perf_test.py:6-29): 1000 one-query requests at
    concurrency 100, k = 10."""
    import ctypes as C
    import numpy as np
    import mse
    from mse import ffi
    H = _callers_lib(root)
    fn = C.cast(ffi.lib().mse_disk_query_topk_f32, C.c_void_p)
    qf = np.ascontiguousarray(queries_f32, np.float32)
    n_all = qf.shape[0]
    checker = mse.Searcher(vecs)
    want_ids, want_sc, _ = mse.disk_query_topk(checker, pq, codes, g, qf, k, None, None, None, disable_pq, beam, search_list)
    checker.close()
    mse.set_coalescer(g, *coalescer)
    # A container may see every core of the host (256 here) and own 16 cores' worth of their TIME (cgroup cpu.max): 4096 request
    # threads spread over 256 mostly idle cores pay for every wake-up with an inter-processor interrupt to a sleeping core and for
    # every queue word with a cache line crossing sockets.  A server is pinned to its allocation; so is this leg: the request threads
    # (and the coalescer's workers, made by the first request) inherit 2 x quota cores (scripts/graph_callers_probe.py ... affinity:
    # 4096 threads 0.36 M queries/s on 256 cores, 0.54 / 0.64 / 0.54 M on 16 / 32 / 64).
    import os
    all_cores, pinned = sorted(os.sched_getaffinity(0)), None
    if pin_to_quota:
        try:
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
            quota = int(round(int(q) / int(per))) if q != "max" else 0
        except Exception:  # noqa: BLE001
            quota = 0
        if quota and len(all_cores) > 4 * quota:
            pinned = all_cores[:2 * quota]
            os.sched_setaffinity(0, pinned)
    n_s = 64
    searchers = [mse.Searcher(vecs) for _ in range(n_s)]        # thread t uses searcher t % 64: a coalesced call only reads its base
    sarr = (C.c_void_p * n_s)(*[s._h for s in searchers])
    pq_h, c_h = (pq._h if pq is not None else None), (codes._h if codes is not None else None)

    def run(T, n):
        ids = np.full((n, k), 0xFFFFFFFF, np.uint32)
        sc = np.zeros((n, k), np.int64)
        lat = np.zeros(n, np.float64)
        failed = C.c_int(0)
        st0 = mse.coalescer_stats(g)
        dt = H.mse_callers_run_query(fn, 1, sarr, n_s, pq_h, c_h, g._h, qf.ctypes.data, n, D * 4, None, 0, int(disable_pq), beam, search_list, k, T,
                                     ids.ctypes.data, sc.ctypes.data, lat.ctypes.data, C.byref(failed))
        st1 = mse.coalescer_stats(g)
        ok = bool(dt > 0 and failed.value == 0 and np.array_equal(ids, want_ids[:n]) and np.array_equal(sc, want_sc[:n]))
        passes = st1["passes"] - st0["passes"]
        busy = (st1["run_us"] - st0["run_us"]) * 1e-6
        return {"threads": T, "queries": n, "queries_per_s": n / dt if dt > 0 else None, "seconds": dt,
                "worker_seconds_in_submissions": busy, "ms_per_submission": busy / max(passes, 1) * 1e3,
                "latency_ms": {"p50": float(np.percentile(lat, 50)), "p99": float(np.percentile(lat, 99)), "max": float(lat.max())},
                "submissions": passes, "queries_per_submission": n / max(passes, 1), "all_answers_equal_the_batch_call": ok,
                "recall_at_10": recall_at(ids, truth[:n]) if truth is not None else None,
                "vs_one_call_of_4096": (n / dt / one_call_qps) if dt > 0 and one_call_qps else None}

    L = ffi.lib()
    afn = [C.cast(getattr(L, nm), C.c_void_p) for nm in ("mse_disk_query_submit_f32", "mse_graph_completions", "mse_ticket_status", "mse_ticket_user", "mse_ticket_free")]
    afn_nocopy = [C.cast(L.mse_disk_query_submit_f32_nocopy, C.c_void_p)] + afn[1:]

    def run_async(W, n, host_threads=1, nocopy=False):
        """the same requests WITHOUT a thread per request: `host_threads` native threads keep W one-query requests in flight as tickets"""
        ids = np.full((n, k), 0xFFFFFFFF, np.uint32)
        sc = np.zeros((n, k), np.int64)
        lat = np.zeros(n, np.float64)
        failed = C.c_int(0)
        st0 = mse.coalescer_stats(g)
        dt = H.mse_callers_run_async(*(afn_nocopy if nocopy else afn), searchers[0]._h, pq_h, c_h, g._h, qf.ctypes.data, n, D * 4, int(disable_pq), beam, search_list, k, W, host_threads,
                                     ids.ctypes.data, sc.ctypes.data, lat.ctypes.data, C.byref(failed))
        st1 = mse.coalescer_stats(g)
        ok = bool(dt > 0 and failed.value == 0 and np.array_equal(ids, want_ids[:n]) and np.array_equal(sc, want_sc[:n]))
        passes = st1["passes"] - st0["passes"]
        busy = (st1["run_us"] - st0["run_us"]) * 1e-6
        return {"in_flight": W, "host_threads": host_threads, "query_copied_at_submit": not nocopy, "queries": n, "queries_per_s": n / dt if dt > 0 else None, "seconds": dt,
                "worker_seconds_in_submissions": busy, "ms_per_submission": busy / max(passes, 1) * 1e3,
                "latency_ms": {"p50": float(np.percentile(lat, 50)), "p99": float(np.percentile(lat, 99)), "max": float(lat.max())},
                "submissions": passes, "queries_per_submission": n / max(passes, 1), "all_answers_equal_the_batch_call": ok,
                "vs_one_call_of_4096": (n / dt / one_call_qps) if dt > 0 and one_call_qps else None}

    points, tickets = [], []
    for T in thread_counts:
        n = min(n_all, T * rounds if rounds else max(10 * T, 20_000))
        run(T, min(n, 2 * T))                       # warm: two rounds
        points.append(run(T, n))
    for W in thread_counts:
        n = min(n_all, W * rounds if rounds else max(10 * W, 20_000))
        run_async(W, min(n, 2 * W))
        tickets.append(run_async(W, n))
    if max(thread_counts) >= 1024:      # the largest window once more with the query left in the caller's buffer (mse_disk_query_submit_f32_nocopy)
        W = max(thread_counts)
        n = min(n_all, W * rounds if rounds else max(10 * W, 20_000))
        tickets.append(run_async(W, n, nocopy=True))
    # (two submitting threads for the largest window measured 0.69-0.83 M against 0.80-1.12 M from one: the submissions get smaller,
    # profiles/r05_graph_callers_tickets.txt -- not part of the leg)
    run(100, min(n_all, 200))
    perf_test = run(100, min(n_all, 1000))
    st = mse.coalescer_stats(g)
    for s in searchers:
        s.close()
    mse.set_coalescer(g, 0, 0, 0)
    if pinned:
        os.sched_setaffinity(0, all_cores)
    return {"metric": "queries/s through the graph index's request path, ONE f32 query per call from T native threads (closed loop), host pointers in and out",
            "request_threads_pinned_to_cores": len(pinned) if pinned else None,
            "search_list": search_list, "beamwidth": beam, "k": k, "neighbours_scored": "exactly" if disable_pq else "by ADC (the reference's default)",
            "points": points,
            "tickets": {"what": "the same one-query requests WITHOUT a thread per request: ONE native thread keeps `in_flight` of them queued as tickets "
                                "(mse_disk_query_submit_f32 / mse_graph_completions), the way the reference's monoio tasks would "
                                "(src/query_disk_index.rs:640-655); latency = submit to collection", "points": tickets},
            "perf_test_py_shape": dict(perf_test, note="1000 one-query requests at concurrency 100, k = 10 (perf_test.py:6-29)"),
            "coalescer": {"max_queries_per_submission": coalescer[0] or 1024, "max_wait_us": coalescer[1] or 200, "workers": coalescer[2] or 3,
                          "submissions_started_by_wait_budget": st["deadline_fires"]}}


HARD_PARAMS = dict(cone=0.62, topic=0.35, within=0.65, noise=0.25, decay=0.6, rank=96)   # scripts/hardness_probe.py config "b"
OOD_GAP, OOD_EXTRA_NOISE = 0.35, 0.30


def train_codec(samp, seed=4, iters=3):
    """OPQ-shaped 64 x 256 codec in aopq_train.py's layout (a rotation + per-subspace max-inner-product k-means), trained on the
    host over a small sample: -> (centroids [256, 1152], transform [1152, 1152])."""
    import numpy as np
    rng = np.random.default_rng(seed)
    T = np.linalg.qr(rng.standard_normal((D, D)))[0].astype(np.float32)
    ts = samp @ T.T
    cents = np.zeros((256, D), np.float32)
    for i in range(64):
        sub = ts[:, i * 18:(i + 1) * 18]
        c = sub[rng.choice(len(sub), 256, replace=False)].copy()
        for _ in range(iters):
            asg = np.argmax(sub @ c.T, axis=1)
            for j in range(256):
                mem = sub[asg == j]
                if len(mem):
                    c[j] = mem.mean(axis=0)
        cents[:, i * 18:(i + 1) * 18] = c
    return cents, T


def train_codec_aopq(samp, queries, rounds=3, iters=300, lr=5e-4, seed=4, kmeans_iters=3):
    """The 64 x 256 codec as diskann/aopq_train.py:33-85 trains it (bench data, not product: the trainer is out of scope, a trained
    codec to evaluate the ADC-scored search with is not): start from a random rotation and per-subspace max-inner-product k-means
    (train_codec), then `rounds` x { `iters` Adam steps on the centroids against the QUERY-AWARE loss E_q[(q . (x' - c(x')))^2] with
    x' = rotated row and c() the codec's own assignment rule (max inner product per subspace, quantize_batch vector.rs:331-364);
    rotation update R = V U^T from the SVD of X^T Y (non-parametric OPQ) }.  Differences from the script, stated: the expectation over
    queries is taken exactly through the queries' second-moment matrix C (loss = sum_x r^T C r, C = E[q' q'^T] over ALL sample
    queries) instead of 2048 sampled queries per step; queries are rotated like the rows (the ADC score is q' . c(x'), :367-405);
    the rows are a sample (torch on the device, seconds).  300 Adam steps per round as in the script (scripts/codec_train_probe.py,
    profiles/r06_codec_train_probe.txt: 3 x 120 / 3 x 300 / 6 x 120 / 10 x 100 steps -> PQ-only recall@10 0.372 / 0.412 / 0.388 / 0.346 on
    the hard set at 1e6 rows: steps per round help, more rotation updates do not).  samp [n, D] / queries [m, D]: float32 cuda tensors.
    -> (centroids [256, D] f32, transform [D, D] f32) in train_codec's layout."""
    import numpy as np
    import torch
    dev = samp.device
    g = torch.Generator(device=dev).manual_seed(seed)
    n = samp.shape[0]
    cents0, T0 = train_codec(samp[torch.randperm(n, device=dev, generator=g)[:20000]].cpu().numpy(), seed=seed, iters=kmeans_iters)
    P = torch.from_numpy(T0.T.copy()).to(dev)                                  # rotated = x @ P
    cent = torch.from_numpy(cents0.reshape(256, 64, 18).transpose(1, 0, 2).copy()).to(dev).requires_grad_(True)   # [64 subspaces][256][18]
    opt = torch.optim.Adam([cent], lr=lr)
    hist = []

    def assign_quant(xr):
        xs = xr.view(-1, 64, 18)
        sims = torch.einsum("nsd,skd->nsk", xs, cent.detach())
        a = sims.argmax(dim=2)                                                # first maximum, as quantize_batch
        q = cent[torch.arange(64, device=dev).unsqueeze(0), a]             # [n, 64, 18], differentiable in the centroids
        return q.reshape(-1, D)

    for rd in range(rounds):
        xr = samp @ P
        qr = queries @ P
        C = (qr.T @ qr) / qr.shape[0]
        ev, U = torch.linalg.eigh(C)
        Lh = U * ev.clamp_min(0).sqrt().unsqueeze(0)                          # C = Lh Lh^T
        for it in range(iters):
            opt.zero_grad(set_to_none=True)
            tot = 0.0
            for i in range(0, n, 32768):
                with torch.enable_grad():      # (a caller may have switched autograd off globally: the towers' code does)
                    r = xr[i:i + 32768] - assign_quant(xr[i:i + 32768])
                    loss = ((r @ Lh) ** 2).sum() / n
                    loss.backward()
                tot += float(loss.detach())
            opt.step()
            if it == 0 or it == iters - 1:
                hist.append(round(tot, 6))
        with torch.no_grad():
            y = assign_quant(xr)
            u, _, vt = torch.linalg.svd(samp.T @ y)                            # X^T Y = U S V^T  ->  R = U V^T minimises |X R - Y|
            P = u @ vt
    cents = cent.detach().permute(1, 0, 2).reshape(256, D).contiguous().cpu().numpy().astype(np.float32)
    # (the centroids were fitted under the previous rotation; one more assignment-consistent k-means pass is NOT run: the script does not either)
    return cents, P.T.contiguous().cpu().numpy().astype(np.float32), {"rounds": rounds, "adam_steps_per_round": iters, "lr": lr,
                                                                        "rows": int(n), "queries": int(queries.shape[0]),
                                                                        "query_aware_loss_first_last_per_round": hist}


def shard_centroid_entries(rows, n_base, n_shards=64, sample=200_000, seed=7):
    """Stand-ins for the index header's shards (centroid + start node each, src/query_disk_index.rs:254-256,447-450) over a one-piece
    index: k-means centroids (spherical, two Lloyd rounds on a row sample) and, per centroid, the sample row closest to it as the
    shard's medioid.  -> (centroids [S, 1152] f32, medioid ids [S] u32)."""
    import numpy as np
    import torch
    g = torch.Generator(device="cuda").manual_seed(seed)
    idx = torch.randint(0, n_base, (min(sample, n_base),), device="cuda", generator=g)
    x = rows[idx].float()
    c = x[torch.randperm(x.shape[0], device="cuda", generator=g)[:n_shards]].clone()
    for _ in range(2):
        a = torch.argmax(x @ c.T, dim=1)
        for j in range(n_shards):
            m = x[a == j]
            if m.shape[0]:
                c[j] = m.mean(dim=0)
    a = torch.argmax(x @ c.T, dim=1)
    sims = x @ c.T
    med = torch.argmax(sims, dim=0)                       # per centroid: the sample row with the largest dot product
    return c.cpu().numpy().astype(np.float32), idx[med].cpu().numpy().astype(np.uint32)


def request_path_leg(s, g, L, K=10, beam=4, n_one=300, cycles=40, in_flight=64, seed=11):
    """The whole request as a user of query_disk_index sees it (src/query_disk_index.rs:345-381 embeds the query text through the clip
    server, :436-540 searches with the embedding): token ids in -> SigLIP text tower -> the f16 row handed to the search ON THE DEVICE
    (mse_siglip_text_encode_dev, mse_searcher_wait_stream: no host round trip between the two) -> entry step, beam search, top-k ->
    k (id, score) pairs on the host.  Measured one request at a time (latency p50 / p99) and with `in_flight` requests at once, the
    way the clip server batches what is waiting: one tower call for all of them, one search call.  Seeded tower weights and synthetic
    tokens (no checkpoint offline): the embedding is a unit vector unrelated to the rows, so a search costs what a query far from the
    data costs (its full search list of node fetches); the arithmetic per request is that of real text."""
    import numpy as np
    import mse
    from mse import siglip
    tcfg = dict(siglip.SO400M_TEXT)
    teng = siglip.SiglipTextEngine.from_state_dict(siglip.synthetic_text_state_dict(tcfg), tcfg, max_batch=in_flight)
    tok = np.random.default_rng(seed).integers(2, tcfg["vocab_size"], size=(max(n_one, in_flight * 4), tcfg["context_length"]), dtype=np.int64)

    def request_dev(t):
        _, p16, stream = teng.encode_text_device(t)
        s.wait_stream(stream)
        return mse.disk_query_topk(s, None, None, g, (p16, t.shape[0]), K, None, None, None, True, beam, L)

    def request_host(t):
        f16 = teng.encode_text(t, out="f16")
        return mse.disk_query_topk(s, None, None, g, f16, K, None, None, None, True, beam, L)

    def pct(ts):
        a = np.sort(np.asarray(ts)) * 1e3
        return {"p50": float(a[len(a) // 2]), "p99": float(a[min(len(a) - 1, int(len(a) * 0.99))]), "mean": float(a.mean()), "n": int(len(a))}

    # the hand-over changes nothing: device path == host round trip, ids and scores
    ids_d, sc_d, _ = request_dev(tok[:8])
    ids_h, sc_h, _ = request_host(tok[:8])
    same = bool(np.array_equal(ids_d, ids_h) and np.array_equal(sc_d, sc_h))
    out = {"what": "token ids in -> SigLIP text tower (27 blocks) -> f16 row handed over on the device -> entry step + beam search + top-k -> host",
           "search_list": int(L), "beamwidth": beam, "k": K, "device_hand_over_equals_host_round_trip": same,
           "query": "the tower's output for synthetic tokens under seeded weights (a unit vector unrelated to the rows)"}
    for fn, key in ((request_dev, "one_at_a_time"), (request_host, "one_at_a_time_host_round_trip")):
        for i in range(5):
            fn(tok[i:i + 1])
        ts = []
        for i in range(n_one):
            t0 = time.perf_counter()
            fn(tok[i:i + 1])
            ts.append(time.perf_counter() - t0)
        out[key] = {"latency_ms": pct(ts), "requests_per_s": len(ts) / float(np.sum(ts))}
    # where one request's time goes: the tower alone (features to the host), the search alone (a host f16 query)
    f16 = teng.encode_text(tok[:1], out="f16")
    ts_t, ts_s = [], []
    for i in range(100):
        t0 = time.perf_counter()
        teng.encode_text(tok[i:i + 1], out="f16")
        ts_t.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        mse.disk_query_topk(s, None, None, g, f16, K, None, None, None, True, beam, L)
        ts_s.append(time.perf_counter() - t0)
    out["one_at_a_time"]["parts_ms"] = {"text_tower_alone_p50": pct(ts_t)["p50"], "search_alone_p50": pct(ts_s)["p50"]}
    # `in_flight` requests at once: one tower call + one search call per cycle, every request of a cycle answered at its end
    for i in range(3):
        request_dev(tok[:in_flight])
    ts = []
    for i in range(cycles):
        b0 = (i % 4) * in_flight
        t0 = time.perf_counter()
        request_dev(tok[b0:b0 + in_flight])
        ts.append(time.perf_counter() - t0)
    out[f"in_flight_{in_flight}"] = {"latency_ms": pct(ts), "requests_per_s": in_flight * len(ts) / float(np.sum(ts)),
                                     "shape": f"{in_flight} requests per cycle: one text-tower call of {in_flight} rows, one search call of {in_flight} device-resident queries"}
    # ... and the same with the searches as TICKETS (mse_disk_query_submit_f32 / completions: the host keeps its one-query requests)
    try:
        tk = mse.QueryTickets(s, None, None, g, K, True, beam, L)
        ts = []
        for i in range(cycles):
            b0 = (i % 4) * in_flight
            t0 = time.perf_counter()
            f32 = teng.encode_text(tok[b0:b0 + in_flight], out="f32")
            for j in range(in_flight):
                tk.submit(f32[j], key=j)
            got = 0
            while got < in_flight:
                got += len(tk.collect(timeout_us=5_000_000))
            ts.append(time.perf_counter() - t0)
        tk.close()
        out[f"in_flight_{in_flight}_tickets"] = {"latency_ms": pct(ts[3:]), "requests_per_s": in_flight * len(ts[3:]) / float(np.sum(ts[3:])),
                                                 "shape": "one text-tower call, features to the host, one TICKET per request, collected as they complete"}
    except Exception as e:  # noqa: BLE001
        out[f"in_flight_{in_flight}_tickets"] = {"error": repr(e)}
    teng.close()
    return out


def graph_index_bench(root, kind, n, batch=16384, passes=1, callers=False, budget_s=None, request_path=False):
    """One row of the graph-index table: a Vamana graph (generate-index-shard's defaults R 64, L 192, C 750; one pass) over n synthetic
    rows of `kind` (easy / hard / ood, see the module docstring), searched through the request path in one call
    (mse_disk_query_topk): operating points picked on 4096 TUNING queries (smallest search list with recall@10 >= 0.96 there) and
    reported on 4096 HELD-OUT queries, for exactly scored neighbours (entry: sampled rows; and the reference's closest-shard-centroid
    rule) and for the reference's default ADC-scored search; the PQ flat scan + fp16 re-rank on the same rows (r picked the same
    way); hardness statistics of the (base, query) pair."""
    import numpy as np
    import torch
    import mse
    K, R, nq_t = 10, 64, 4096
    n_qtrain = 100_000 if kind == "ood" else 0
    t_all = time.perf_counter()
    if kind == "easy":
        gen = easy_generator(n)
        rows = gen(n, 1)
        tune_q, held_q = gen(nq_t, 2), gen(nq_t, 3)
    else:
        hs = HardSet(n, **HARD_PARAMS)
        rows = torch.empty(n + n_qtrain, D, device="cuda", dtype=torch.float16)
        rows[:n] = hs.rows(n, 1)
        if kind == "ood":
            rows[n:] = hs.rows(n_qtrain, 4, queries="ood", gap=OOD_GAP, extra_noise=OOD_EXTRA_NOISE)      # the query sample the graph is built with
            tune_q = hs.rows(nq_t, 2, queries="ood", gap=OOD_GAP, extra_noise=OOD_EXTRA_NOISE)
            held_q = hs.rows(nq_t, 3, queries="ood", gap=OOD_GAP, extra_noise=OOD_EXTRA_NOISE)
        else:
            tune_q, held_q = hs.rows(nq_t, 2), hs.rows(nq_t, 3)
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t_all
    n_all = n + n_qtrain
    vecs = mse.VectorList.wrap_device(rows.data_ptr(), n_all, D, keepalive=rows)         # what the graph is built over (base rows, then the query sample)
    base_only = mse.VectorList.wrap_device(rows.data_ptr(), n, D, keepalive=rows) if n_qtrain else vecs
    s = mse.Searcher(vecs)
    sb = mse.Searcher(base_only) if n_qtrain else s
    out = {"kind": kind, "rows": n, "generated_on_device_seconds": t_gen}
    # exact answers + hardness
    hard_t, truth_t = hardness(base_only, sb, rows[:n], tune_q)
    _, truth_h = sb.bruteforce_topk(held_q.cpu().numpy().view(np.uint16), K)
    out["hardness"] = hard_t
    # build
    t0 = time.perf_counter()
    med = mse.medioid(vecs)
    g = mse.BuildGraph(n_all, R)
    g.random_fill(1)
    order = np.random.default_rng(3).permutation(n_all).astype(np.uint32)
    cfg = mse.IndexBuildConfig(r=R, l=192, maxc=750, query_breakpoint=n if n_qtrain else 0xFFFFFFFF)
    for _ in range(passes):
        g.build(s, order, med, cfg, batch)
    t_pass = time.perf_counter() - t0
    if n_qtrain:
        q_order = (n + np.random.default_rng(4).permutation(n_qtrain)).astype(np.uint32)
        g.robust_stitch(s, q_order, cfg)
    t_build = time.perf_counter() - t0
    out["build"] = {"seconds": t_build, "points_per_s": n_all * passes / t_pass, "passes": passes, "r": R, "l": 192, "maxc": 750, "batch": batch,
                    "query_sample": n_qtrain, "robust_stitch_seconds": (t_build - t_pass) if n_qtrain else None}
    qt16, qh16 = tune_q.cpu().numpy().view(np.uint16), held_q.cpu().numpy().view(np.uint16)
    qt32, qh32 = tune_q.float().cpu().numpy(), held_q.float().cpu().numpy()

    def pick(run, grid, goal=0.96):
        """smallest grid value whose TUNING recall reaches the goal (when none does: the one with the best tuning recall, labelled); the
        held-out point is measured once, warm"""
        sweep, at, best = [], None, None
        for v in grid:
            top = run(v, qt16, qt32)[0]
            rec = recall_at(top, truth_t)
            sweep.append([v, round(rec, 4)])
            if best is None or rec > best[1]:
                best = (v, rec)
            if rec >= goal:
                at = v
                break
        v = at if at is not None else best[0]
        run(v, qh16, qh32)
        t0 = time.perf_counter()
        top, extra = run(v, qh16, qh32)
        dt = time.perf_counter() - t0
        chosen = dict({"value": v, "queries_per_s": nq_t / dt, "recall_at_10": recall_at(top, truth_h), "queries": nq_t, "goal_reached": at is not None}, **extra)
        return {"tuning_sweep": sweep, "held_out": chosen, "tuning_goal": goal}

    grid_L = (12, 16, 24, 32, 48, 64, 100, 150, 200, 300, 400, 500, 600, 800)
    # (1) exactly scored neighbours, entry = the sampled row with the largest dot product
    n_entry = max(4096, n // 1500)
    e_idx = np.sort(np.random.default_rng(5).choice(n, n_entry, replace=False)).astype(np.uint32)
    mse.set_entries(g, vecs, e_idx)

    def run_exact(L, q16, q32):
        top, _, st = mse.disk_query_topk(s, None, None, g, q16, K, None, None, None, True, 4, L)
        return top, {"node_fetches_per_query": float(st["cmps"].mean())}
    out["exact_scored"] = dict(pick(run_exact, grid_L), entry=f"{n_entry} sampled rows, exact top-1 (timed)", beamwidth=4)
    L_exact = (out["exact_scored"]["held_out"] or {}).get("value")
    # the search kernel against the HBM roofline: ALGORITHMIC bytes = what the searches gathered (counted by the kernel itself: every
    # exactly scored row is one 2304-byte gather, every fetched node one adjacency list of 64 ids + its degree) / the kernel's duration by
    # HIP events on the searcher's stream (mse_searcher_beam_timing); profiles/r06_beam_search_hard_pmc.txt holds the rocprofv3 view
    if L_exact:
        try:
            s.beam_timing(2)
            for _ in range(3):
                run_exact(L_exact, qh16, qh32)
            m = s.beam_timing(0)
            row_b, adj_b = m["rows_scored"] * D * 2, m["nodes_fetched"] * (R * 4 + 4)
            gbps = (row_b + adj_b) / (m["kernel_ms"] * 1e-3) / 1e9
            out["gather_roofline"] = {"bound": "hbm", "kernel": "beam_search_kernel<64> (one wave per query)" if nq_t > 1024 and L_exact <= 256 else "beam_search_kernel<256>",
                                      "achieved": gbps, "peak": 8000.0, "unit": "GB/s", "frac": gbps / 8000.0,
                                      "bytes_per_launch": (row_b + adj_b) / m["launches"], "avg_launch_ms": m["kernel_ms"] / m["launches"], "launches_timed": m["launches"],
                                      "queries_per_launch": m["queries"] / m["launches"], "kernel_queries_per_s": m["queries"] / (m["kernel_ms"] * 1e-3),
                                      "rows_scored_per_query": m["rows_scored"] / m["queries"], "nodes_fetched_per_query": m["nodes_fetched"] / m["queries"],
                                      "bytes_per_query": (row_b + adj_b) / m["queries"],
                                      "algorithmic_bytes": "rows scored exactly x 2304 B + fetched nodes x 260 B (64 neighbour ids + the degree), counted by the kernel"}
        except Exception as e:  # noqa: BLE001
            out["gather_roofline"] = {"error": repr(e)}
    # the same search list at other beam widths (the server's `beam_width` is the operator's, query_disk_index.rs:63,452; `evaluate` uses 3):
    # held-out queries, one call each -- a narrower beam is more iterations of less work, which 4096 concurrent searches hide
    if L_exact:
        widths = {}
        for bw in (1, 2, 8):
            mse.disk_query_topk(s, None, None, g, qh16, K, None, None, None, True, bw, L_exact)
            t0 = time.perf_counter()
            top_b, _, _ = mse.disk_query_topk(s, None, None, g, qh16, K, None, None, None, True, bw, L_exact)
            widths[str(bw)] = [nq_t / (time.perf_counter() - t0), recall_at(top_b, truth_h)]
        out["exact_scored"]["other_beam_widths_same_list"] = dict(widths, columns="[queries/s, recall@10 held out]")
        # ... and the operating point with the beam width as a second knob: the fastest of 4 / 2 / 1 on the TUNING queries that meets the goal
        tune_b, best_b = [], (4, None)
        for bw in (4, 2, 1):
            mse.disk_query_topk(s, None, None, g, qt16, K, None, None, None, True, bw, L_exact)
            t0 = time.perf_counter()
            top_b, _, _ = mse.disk_query_topk(s, None, None, g, qt16, K, None, None, None, True, bw, L_exact)
            qps_b, rec_b = nq_t / (time.perf_counter() - t0), recall_at(top_b, truth_t)
            tune_b.append([bw, round(qps_b, 1), round(rec_b, 4)])
            if bw == 4 or (rec_b >= 0.96 and qps_b > 1.05 * best_b[1]):     # (a narrower beam has to win by more than the timing noise)
                best_b = (bw, qps_b)
        ho = out["exact_scored"]["held_out"]
        pt_b = [ho["queries_per_s"], ho["recall_at_10"]] if best_b[0] == 4 else widths[str(best_b[0])]
        out["exact_scored_best_beam"] = {"beamwidth": best_b[0], "value": L_exact, "queries_per_s": pt_b[0], "recall_at_10": pt_b[1],
                                         "tuning": tune_b, "columns": "[beam, queries/s, recall@10] on the tuning queries; the point itself on the held-out ones"}
    # (2) the same with the reference's entry rule: closest shard centroid -> that shard's medioid
    cen, med_ids = shard_centroid_entries(rows, n)
    mse.set_entry_centroids(g, cen, med_ids)

    def run_cen(L, q16, q32):
        top, _, st = mse.disk_query_topk(s, None, None, g, q32, K, None, None, None, True, 4, L)
        return top, {"node_fetches_per_query": float(st["cmps"].mean())}
    out["exact_scored_reference_entry_rule"] = dict(pick(run_cen, grid_L), entry=f"{len(med_ids)} shard centroids (k-means of a row sample) -> the shard's medioid; "
                                                    "scale_dot_result_f64(dot(centroid, query)), last maximum (src/query_disk_index.rs:447-450)", beamwidth=4)
    # (3) the reference's default: neighbours scored by ADC (64 x 8-bit OPQ codes, 64 KiB table per query in LDS), fetched nodes exactly
    t0 = time.perf_counter()
    sel = torch.from_numpy(np.sort(np.random.default_rng(4).choice(n, min(n, 100_000), replace=False))).cuda()
    # training queries: from the distribution the index will be asked from, none of them a tuning or held-out query
    if kind == "ood":
        train_q = rows[n:n + 50_000].float()
    elif kind == "easy":
        train_q = gen(50_000, 7).float()
    else:
        train_q = hs.rows(50_000, 7).float()
    codec_info = None
    try:
        cents, T, codec_info = train_codec_aopq(rows[sel].float(), train_q)
    except Exception as e:  # noqa: BLE001 -- (no torch.linalg on this build, out of memory, ...): the starting point alone
        codec_info = {"error": repr(e), "fallback": "random rotation + max-inner-product k-means (the trainer's starting point)"}
        cents, T = train_codec(rows[sel[:20000]].float().cpu().numpy())
    del train_q
    t_codec = time.perf_counter() - t0
    pq = mse.ProductQuantizer(cents, T, 18, D)
    t0 = time.perf_counter()
    codes = mse.Codes.quantize_base(pq, vecs)
    t_quant = time.perf_counter() - t0
    mse.set_entries(g, vecs, e_idx)

    def run_adc(L, q16, q32):
        top, _, st = mse.disk_query_topk(s, pq, codes, g, q32, K, None, None, None, False, 4, L)
        return top, {"node_fetches_per_query": float(st["cmps"].mean()), "adc_scores_per_query": float(st["pq_cmps"].mean())}
    try:
        out["adc_scored"] = dict(pick(run_adc, grid_L), entry=f"{n_entry} sampled rows", beamwidth=4,
                                 note="query_disk_index's default (disable_pq = false, :195-207): neighbours enter the list by their ADC score, "
                                      "fetched nodes are scored exactly; f32 queries in, tables made on the device")
    except Exception as e:  # noqa: BLE001
        out["adc_scored"] = {"error": repr(e)}
    # (4) configs[4] on the same rows: flat ADC scan of all codes, top-r, exact fp16 re-rank, top-10; and the ADC-only recall
    bcodes = codes if not n_qtrain else mse.Codes.quantize_base(pq, base_only)

    def run_pq(r, q16, q32):
        tops = [pq.scan_topk_batch(bcodes, q32[i:i + 64], r, K, sb)[1] for i in range(0, q32.shape[0], 64)]
        return np.concatenate(tops), {"queries_per_call": 64}
    try:
        out["pq_rerank"] = pick(run_pq, (50, 100, 200, 400, 800, 1600))
        adc_only = np.concatenate([pq.scan_topk_batch(bcodes, qt32[i:i + 64], K, K, None)[1] for i in range(0, 1024, 64)])
        out["pq_only_recall_at_10"] = recall_at(adc_only, truth_t[:1024])
        out["codes"] = {"made_on_device_seconds": t_quant, "vectors_per_s": n_all / t_quant, "codec_trained_seconds": t_codec,
                        "codec": "64 x 256, trained as diskann/aopq_train.py:33-85 trains it (bench-side, torch on the device, outside every timed "
                                 "region): random rotation + max-inner-product k-means, then rounds of Adam steps on the centroids against the "
                                 "query-aware loss E_q[(q . residual)^2] (3 x 300 steps) and SVD rotation updates, on a 100 000-row sample with 50 000 training queries",
                        "training": codec_info}
        # the trainer's STARTING point on the same rows, for the difference the training makes (rounds 1-5 reported this codec)
        c0, T0 = train_codec(rows[sel[:20000]].float().cpu().numpy())
        pq0 = mse.ProductQuantizer(c0, T0, 18, D)
        codes0 = mse.Codes.quantize_base(pq0, base_only)
        adc0 = np.concatenate([pq0.scan_topk_batch(codes0, qt32[i:i + 64], K, K, None)[1] for i in range(0, 1024, 64)])
        out["pq_only_recall_at_10_untrained_codec"] = recall_at(adc0, truth_t[:1024])
        del codes0, pq0
    except Exception as e:  # noqa: BLE001
        out["pq_rerank"] = {"error": repr(e)}
    # (5) the request path in the reference's call shape, at the exact-scored operating point
    if callers and L_exact:
        try:
            one_call = out["exact_scored"]["held_out"]["queries_per_s"]
            call_q = (gen(40960, 6) if kind == "easy" else hs.rows(40960, 6, queries="ood", gap=OOD_GAP, extra_noise=OOD_EXTRA_NOISE) if kind == "ood"
                      else hs.rows(40960, 6))
            _, truth_c = sb.bruteforce_topk(call_q.cpu().numpy().view(np.uint16), K)
            out["graph_callers"] = graph_callers(root, vecs, g, call_q.float().cpu().numpy(), truth_c, L_exact, K, 4, (64, 512, 4096), one_call)
        except Exception as e:  # noqa: BLE001
            out["graph_callers"] = {"error": repr(e)}
    # (6) the whole request: text in -> top-k out, the tower's output handed to the search on the device
    if request_path and L_exact:
        try:
            mse.set_entries(g, vecs, e_idx)
            out["request_path"] = request_path_leg(s, g, L_exact, K, 4)
        except Exception as e:  # noqa: BLE001
            out["request_path"] = {"error": repr(e)}
    out["seconds"] = time.perf_counter() - t_all
    g.close()
    return out


# ---- the approximate-search paths over several GPUs (bench.py --gpus N) ---------------------------------------------------------
ANN_SEED = 0x5EED0011


def _random_codec(seed=0):
    import numpy as np
    rng = np.random.default_rng(seed)
    cents = (rng.standard_normal((256, D)) / np.sqrt(D)).astype(np.float32)
    T = np.linalg.qr(rng.standard_normal((D, D)))[0].astype(np.float32)
    return cents, T


def _shard_ann_state(mse, vecs, searcher, n_g, device, cents, T, graph_list=64, batch=4096):
    """codec, codes and a one-pass Vamana graph (with a sampled entry table) over one shard's rows, on the shard's device"""
    import numpy as np
    from mse import ffi
    ffi.check(ffi.lib().mse_set_device(device), "mse_set_device")
    pq = mse.ProductQuantizer(cents, T, 18, D)
    codes = mse.Codes.quantize_base(pq, vecs)
    g = mse.BuildGraph(n_g, 64)
    g.random_fill(1)
    order = np.random.default_rng(3).permutation(n_g).astype(np.uint32)
    g.build(searcher, order, int(order[0]), mse.IndexBuildConfig(r=64, l=graph_list, maxc=300), batch)
    entries = np.sort(np.random.default_rng(5).choice(n_g, max(256, n_g // 1500), replace=False)).astype(np.uint32)
    mse.set_entries(g, vecs, entries)
    return pq, codes, g


def sharded_ann_inprocess(n_gpus, n_dev, rows_per_gpu=2_000_000, k=10, r=200, search_list=32):
    """`python bench.py --gpus N` (one process, a host thread per shard): the PQ flat scan + exact re-rank and the graph index over
    N shards of `rows_per_gpu` rows each -- codes, descriptors-free, one graph per shard -- through mse_shard_group_pq_scan_topk /
    mse_shard_group_query_topk (the brute-force path's exchange: RCCL all-gather of the packed blocks when every shard has its own
    device, peer stores otherwise).  Checked in the run: the sharded PQ answer against the UNSHARDED call over a full copy of the rows
    on device 0 (bit for bit), the sharded graph answer against the host merge of the per-shard calls."""
    import threading
    import numpy as np
    import mse
    from mse import ffi, shard
    n = rows_per_gpu * n_gpus
    devs = [g % n_dev for g in range(n_gpus)]
    grp = mse.ShardGroup(n_gpus, D, devices=devs)
    grp.generate(ANN_SEED, 0, n)
    exchange = "peer stores"
    if len(set(devs)) == n_gpus:
        try:
            grp.set_exchange(grp.EXCHANGE_RCCL)
            exchange = f"ONE ncclAllGather per exchange, {grp.rccl_ranks} ranks"
        except mse.MseError as e:
            exchange = f"peer stores (RCCL unavailable: {e})"
    cents, T = _random_codec()
    state, errs = [None] * n_gpus, []

    def make(gi):
        try:
            state[gi] = _shard_ann_state(mse, grp.base(gi), grp.searcher(gi), len(grp.base(gi)), devs[gi], cents, T)
        except Exception as e:  # noqa: BLE001
            errs.append(f"shard {gi}: {e!r}")

    t0 = time.perf_counter()
    th = [threading.Thread(target=make, args=(gi,)) for gi in range(n_gpus)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    ffi.check(ffi.lib().mse_set_device(devs[0]), "mse_set_device")
    if errs:
        grp.close()
        return {"error": "; ".join(errs)}
    t_prep = time.perf_counter() - t0
    for gi, (pq, codes, g) in enumerate(state):
        grp.attach_pq(gi, pq, codes)
        grp.attach_graph(gi, g)
    rng = np.random.default_rng(9)
    q32 = rng.standard_normal((1024, D)).astype(np.float32)
    q32 /= np.linalg.norm(q32, axis=1, keepdims=True)
    q16 = q32.astype(np.float16).view(np.uint16)
    out = {"shards": n_gpus, "devices": devs, "rows_per_gpu": rows_per_gpu, "rows": n, "exchange": exchange, "prepare_seconds": t_prep,
           "prepare": "per shard, in parallel: 64 x 8-bit codes of the resident rows, a one-pass Vamana graph (R 64, L 64), a sampled entry table"}
    # PQ scan + exact re-rank, 32 queries per call
    grp.pq_scan_topk(q32[:32], r, k)
    t0, calls = time.perf_counter(), 0
    while calls < 10 or time.perf_counter() - t0 < 1.0:
        ps, pi = grp.pq_scan_topk(q32[32 * (calls % 8):32 * (calls % 8) + 32], r, k)
        calls += 1
    dt = time.perf_counter() - t0
    out["pq_scan_rerank"] = {"queries_per_s": 32 * calls / dt, "ms_per_call_of_32": dt / calls * 1e3, "r": r, "k": k, "timing_ms": grp.last_timing(),
                             "exchanges_per_call": 2, "bytes_per_rank_per_exchange": int(ffi.lib().mse_topk_block_bytes(32, r))}
    # the graph index, 1024 queries per call, neighbours scored exactly
    grp.query_topk(q16, k, None, None, True, 4, search_list)
    t0, calls = time.perf_counter(), 0
    while calls < 5 or time.perf_counter() - t0 < 1.0:
        gs, gi_ = grp.query_topk(q16, k, None, None, True, 4, search_list)
        calls += 1
    dt = time.perf_counter() - t0
    out["graph_index"] = {"queries_per_s": 1024 * calls / dt, "ms_per_call_of_1024": dt / calls * 1e3, "search_list": search_list, "beamwidth": 4,
                          "timing_ms": grp.last_timing(), "bytes_per_rank_per_exchange": int(ffi.lib().mse_topk_block_bytes(1024, k))}
    # checks
    try:
        parts_s, parts_i = [], []
        for gi, (pq, codes, g) in enumerate(state):
            ffi.check(ffi.lib().mse_set_device(devs[gi]), "mse_set_device")
            s_g = mse.Searcher(grp.base(gi))
            ids, sc, _ = mse.disk_query_topk(s_g, None, None, g, q16, k, None, None, None, True, 4, search_list)
            parts_s.append(sc)
            parts_i.append(np.where(ids == 0xFFFFFFFF, ids, ids + np.uint32(grp.first_row(gi))))
            s_g.close()
        ffi.check(ffi.lib().mse_set_device(devs[0]), "mse_set_device")
        ws, wi = shard.merge_topk_numpy(np.concatenate(parts_s, 1), np.concatenate(parts_i, 1), k)
        out["graph_index"]["equals_the_merge_of_per_shard_calls"] = bool(np.array_equal(gs, ws) and np.array_equal(gi_, wi))
        free_b, total_b = ffi.sz(), ffi.sz()
        ffi.check(ffi.lib().mse_device_mem_info(free_b, total_b))
        if n * (D * 2 + 64) + (4 << 30) < free_b.value:
            full = mse.VectorList.generate(ANN_SEED, 0, n, D)
            pq0 = mse.ProductQuantizer(cents, T, 18, D)
            codes0 = mse.Codes.quantize_base(pq0, full)
            s0 = mse.Searcher(full)
            us, ui = pq0.scan_topk_batch(codes0, q32[:32], r, k, s0)
            ps, pi = grp.pq_scan_topk(q32[:32], r, k)
            out["pq_scan_rerank"]["equals_the_unsharded_call_bit_for_bit"] = bool(np.array_equal(ps, us) and np.array_equal(pi, ui))
            s0.close(); codes0.close(); full.close()
        else:
            out["pq_scan_rerank"]["equals_the_unsharded_call_bit_for_bit"] = "not checked: a full copy of the rows does not fit beside the index on device 0"
    except Exception as e:  # noqa: BLE001
        out["check_error"] = repr(e)
    grp.close()
    return out


def sharded_ann_rank(comm, dist, rank, world, rows_per_gpu=2_000_000, k=10, r=200, search_list=32):
    """torchrun's shape (one process per GPU): this rank's shard of the same index, mse_comm_pq_scan_topk / mse_comm_query_topk; rank 0
    also checks the PQ answer against the unsharded call over a full copy of the rows."""
    import numpy as np
    import torch
    import mse
    from mse import ffi
    n = rows_per_gpu * world
    lo, hi = rank * rows_per_gpu, (rank + 1) * rows_per_gpu
    vecs = mse.VectorList.generate(ANN_SEED, lo, hi - lo, D)
    s = mse.Searcher(vecs)
    cents, T = _random_codec()
    t0 = time.perf_counter()
    pq, codes, g = _shard_ann_state(mse, vecs, s, hi - lo, torch.cuda.current_device(), cents, T)
    dist.barrier()
    t_prep = time.perf_counter() - t0
    rng = np.random.default_rng(9)
    q32 = rng.standard_normal((1024, D)).astype(np.float32)
    q32 /= np.linalg.norm(q32, axis=1, keepdims=True)
    q16 = q32.astype(np.float16).view(np.uint16)
    out_s = torch.empty((1024, k), dtype=torch.int64, device="cuda")
    out_i = torch.empty((1024, k), dtype=torch.int32, device="cuda")
    res = {"ranks": world, "rows_per_gpu": rows_per_gpu, "rows": n, "prepare_seconds": t_prep, "exchange": "ONE ncclAllGather per exchange (mse_comm)"}

    def timed(fn, per_call, min_calls):
        fn(0)
        dist.barrier()
        t0, calls = time.perf_counter(), 0
        while calls < min_calls:
            fn(calls)
            calls += 1
        torch.cuda.synchronize()
        dist.barrier()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return per_call * calls / float(t.item()), float(t.item()) / calls * 1e3

    qps, ms = timed(lambda c: comm.pq_scan_topk(pq, codes, s, q32[32 * (c % 8):32 * (c % 8) + 32], r, k, lo, out_s.data_ptr(), out_i.data_ptr()), 32, 40)
    res["pq_scan_rerank"] = {"queries_per_s": qps, "ms_per_call_of_32": ms, "r": r, "k": k, "exchanges_per_call": 2}
    qps, ms = timed(lambda c: comm.query_topk(s, g, q16, k, lo, out_s.data_ptr(), out_i.data_ptr(), disable_pq=True, beamwidth=4, search_list=search_list), 1024, 20)
    res["graph_index"] = {"queries_per_s": qps, "ms_per_call_of_1024": ms, "search_list": search_list, "beamwidth": 4}
    comm.pq_scan_topk(pq, codes, s, q32[:32], r, k, lo, out_s.data_ptr(), out_i.data_ptr())
    ps, pi = out_s[:32].cpu().numpy(), out_i[:32].cpu().numpy().view(np.uint32)
    if rank == 0:
        try:
            free_b, total_b = ffi.sz(), ffi.sz()
            ffi.check(ffi.lib().mse_device_mem_info(free_b, total_b))
            if n * (D * 2 + 64) + (4 << 30) < free_b.value:
                full = mse.VectorList.generate(ANN_SEED, 0, n, D)
                codes0 = mse.Codes.quantize_base(pq, full)
                s0 = mse.Searcher(full)
                us, ui = pq.scan_topk_batch(codes0, q32[:32], r, k, s0)
                res["pq_scan_rerank"]["equals_the_unsharded_call_bit_for_bit"] = bool(np.array_equal(ps, us) and np.array_equal(pi, ui))
                s0.close(); codes0.close(); full.close()
        except Exception as e:  # noqa: BLE001
            res["check_error"] = repr(e)
    dist.barrier()
    g.close(); codes.close(); s.close(); vecs.close()
    return res


def graph_index_1e8(root, rate_1e7_points_per_s, budget_s=1000.0, n=100_000_000, batch=16384, kind="easy", max_passes=2):
    """The graph index AT THE METRIC'S SIZE under the command's own clock: ONE Vamana graph over 1e8 x 1152 easy-set rows (230 GB of
    rows + 26 GB of graph in the 288 GB of one MI355X), one pass (generate-index-shard's default), searched through the request path
    in one call; operating point on 4096 tuning queries, reported on 4096 held-out ones.  Guarded by time: the build is predicted
    from the 1e7-row build rate of the SAME run (a 1e8-row pass runs at about 0.75 of it: the rows no longer fit the caches' reach)
    and skipped, with the prediction as the reason, when it would not fit `budget_s`."""
    import numpy as np
    import torch
    import mse
    from mse import ffi
    if not rate_1e7_points_per_s:
        return {"skipped": "no 1e7-row build rate measured in this run to predict the build from"}
    # (measured twice in round 5: the first 1e8-row pass ran at 0.84 and 0.96 of the same run's 1e7-row rate, the second pass at 0.96)
    per_pass = n / (0.8 * rate_1e7_points_per_s)
    predicted = per_pass
    if predicted + 30 > budget_s:
        return {"skipped": f"a one-pass build of {n:.0e} rows is predicted to take {predicted:.0f} s (0.75 x the {rate_1e7_points_per_s:.0f} points/s "
                           f"measured at 1e7 rows in this run) against a budget of {budget_s:.0f} s", "predicted_build_seconds": predicted}
    free_b, total_b = ffi.sz(), ffi.sz()
    ffi.check(ffi.lib().mse_device_mem_info(free_b, total_b))
    need = n * D * 2 + n * 65 * 4 + (14 << 30)
    if need > free_b.value:
        return {"skipped": f"{n} rows + graph + build scratch need {need / 1e9:.0f} GB, {free_b.value / 1e9:.0f} GB free"}
    K, R, nq_t = 10, 64, 4096
    t_all = time.perf_counter()
    if kind == "hard":      # round 6: the set the metric deserves (no micro-clusters, relative contrast 3.1), as graph_index_bench's
        hs = HardSet(n, **HARD_PARAMS)
        rows = hs.rows(n, 1)
        tune_q, held_q = hs.rows(nq_t, 2), hs.rows(nq_t, 3)
    else:
        gen = easy_generator(n)
        rows = gen(n, 1)
        tune_q, held_q = gen(nq_t, 2), gen(nq_t, 3)
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t_all
    vecs = mse.VectorList.wrap_device(rows.data_ptr(), n, D, keepalive=rows)
    s = mse.Searcher(vecs)
    qt16, qh16 = tune_q.cpu().numpy().view(np.uint16), held_q.cpu().numpy().view(np.uint16)
    t0 = time.perf_counter()
    _, truth_t = s.bruteforce_topk(qt16, K)
    _, truth_h = s.bruteforce_topk(qh16, K)
    t_exact = time.perf_counter() - t0
    t0 = time.perf_counter()
    med = mse.medioid(vecs)
    g = mse.BuildGraph(n, R)
    g.random_fill(1)
    order = np.random.default_rng(3).permutation(n).astype(np.uint32)
    pass_s = []
    # generate-index-shard's second pass (-s) when it still fits once the first one has been TIMED (a one-pass graph of this size tops
    # out at recall@10 0.96; the second pass is no slower than the first: the searches start from a better graph)
    while len(pass_s) < max_passes:
        if pass_s and (time.perf_counter() - t_all) + pass_s[0] + 60 > budget_s:
            break
        tp = time.perf_counter()
        g.build(s, order, med, mse.IndexBuildConfig(r=R, l=192, maxc=750), batch)
        pass_s.append(time.perf_counter() - tp)
    passes = len(pass_s)
    t_build = time.perf_counter() - t0
    n_entry = n // 1500
    e_idx = np.sort(np.random.default_rng(5).choice(n, n_entry, replace=False)).astype(np.uint32)
    mse.set_entries(g, vecs, e_idx)
    sweep, chosen, best = [], None, None
    goal = (0.96 if kind == "hard" else 0.97) if passes > 1 else 0.955
    for L in ((100, 200, 400, 600, 800, 1024) if kind == "hard" else (12, 16, 24, 32, 48, 64, 100, 200, 400)):
        mse.disk_query_topk(s, None, None, g, qt16, K, None, None, None, True, 4, L)
        t0 = time.perf_counter()
        top, _, _ = mse.disk_query_topk(s, None, None, g, qt16, K, None, None, None, True, 4, L)
        dt_l = time.perf_counter() - t0
        rec = recall_at(top, truth_t)
        sweep.append([L, round(rec, 4), round(nq_t / dt_l, 1)])
        if best is None or rec > best[1]:
            best = (L, rec)
        if rec >= goal:
            chosen = L
            break
    L = chosen or best[0]
    # beam width (the operator's parameter, query_disk_index.rs:63,452) at that search list, on the TUNING queries: the fastest of 4 / 2 / 1
    # that still meets the tuning goal -- a narrower beam is more iterations of less work, which 4096 concurrent searches hide
    beam, beam_sweep, best_t = 4, [], None
    for bw in (4, 2, 1):
        mse.disk_query_topk(s, None, None, g, qt16, K, None, None, None, True, bw, L)
        t0 = time.perf_counter()
        top_b, _, _ = mse.disk_query_topk(s, None, None, g, qt16, K, None, None, None, True, bw, L)
        qps_b, rec_b = nq_t / (time.perf_counter() - t0), recall_at(top_b, truth_t)
        beam_sweep.append([bw, round(qps_b, 1), round(rec_b, 4)])
        if bw == 4 or (rec_b >= goal and qps_b > 1.05 * best_t):     # (a narrower beam has to win by more than the timing noise)
            beam, best_t = bw, qps_b
    mse.disk_query_topk(s, None, None, g, qh16, K, None, None, None, True, beam, L)
    t0 = time.perf_counter()
    top, _, st = mse.disk_query_topk(s, None, None, g, qh16, K, None, None, None, True, beam, L)
    dt = time.perf_counter() - t0
    s.beam_timing(2)
    mse.disk_query_topk(s, None, None, g, qh16, K, None, None, None, True, 4, L)
    t0 = time.perf_counter()
    top4, _, _ = mse.disk_query_topk(s, None, None, g, qh16, K, None, None, None, True, 4, L)
    beam4 = [nq_t / (time.perf_counter() - t0), recall_at(top4, truth_h)]
    m = s.beam_timing(0)
    gather = None
    if m["launches"]:
        gb = (m["rows_scored"] * D * 2 + m["nodes_fetched"] * (R * 4 + 4)) / (m["kernel_ms"] * 1e-3) / 1e9
        gather = {"bound": "hbm", "achieved": gb, "peak": 8000.0, "unit": "GB/s", "frac": gb / 8000.0, "avg_launch_ms": m["kernel_ms"] / m["launches"],
                  "rows_scored_per_query": m["rows_scored"] / m["queries"], "nodes_fetched_per_query": m["nodes_fetched"] / m["queries"], "at": "beam 4, the chosen search list"}
    out = {"metric": "queries/sec over a 1e8x1152 graph index @ recall@10>=0.95 (ONE Vamana graph, %d pass%s, GPU-resident beam search)" % (passes, "es" if passes > 1 else ""),
           "value": nq_t / dt, "unit": "queries/s", "recall_at_10": recall_at(top, truth_h), "search_list": L, "beamwidth": beam, "queries": nq_t,
           "beam_width_tuning": {"rule": "the fastest of beam 4 / 2 / 1 on the tuning queries that meets the tuning goal at the chosen search list",
                                 "sweep": beam_sweep, "columns": "[beam, queries/s, recall@10] on the tuning queries",
                                 "held_out_at_beam_4": beam4},
           "operating_point": ("smallest search list with tuning recall >= %s" % goal) if chosen else "no search list reached the tuning goal %s: the best one" % goal,
           "tuning_sweep": sweep, "tuning_sweep_columns": "[search list, recall@10 on the tuning queries, queries/s of the second call at that list]", "set": kind, "gather_roofline": gather, "node_fetches_per_query": float(st["cmps"].mean()),
           "build": {"seconds": t_build, "points_per_s": n * passes / sum(pass_s), "passes": passes, "seconds_per_pass": pass_s, "r": R, "l": 192, "maxc": 750,
                     "batch": batch, "predicted_seconds": predicted},
           "entry": f"{n_entry} sampled rows, exact top-1 (timed)", "exact_scan_same_rows_queries_per_s": 2 * nq_t / t_exact,
           "config": {"workload": f"{n} x {D} fp16 {kind}-set rows generated on the device in {t_gen:.1f} s; host arrays in and out"},
           "seconds": time.perf_counter() - t_all}
    g.close()
    s.close()
    return out
