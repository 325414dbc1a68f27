"""The reference's call shape -- ONE query per request from many threads (src/main.rs:896-934,1043-1049;
src/query_disk_index.rs:711-736) -- through the cross-thread coalescer behind the C ABI (csrc/dispatch.hip): every caller must
get exactly the oracle's answer for ITS query, whatever it was batched with."""
import threading

import numpy as np
import pytest

from conftest import SEED_BASE, SEED_QUERY, make_pq

pytestmark = pytest.mark.gpu
D = 1152


def run_threads(n, fn):
    """fn(i) on n threads released together; exceptions are re-raised in the caller."""
    errs, out = [None] * n, [None] * n
    gate = threading.Barrier(n)

    def body(i):
        try:
            gate.wait()
            out[i] = fn(i)
        except BaseException as e:  # noqa: BLE001
            errs[i] = e

    ts = [threading.Thread(target=body, args=(i,)) for i in range(n)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for e in errs:
        if e is not None:
            raise e
    return out


def test_64_threads_one_query_each_get_their_own_oracle_answer(gpu, mse, orc):
    """64 threads x nq = 1, mixed k, three rounds each, through an explicit dispatcher; the passes really are shared."""
    n, T = 20000, 64
    base = orc.gen_rows_f16(SEED_BASE, 0, n)
    q = orc.gen_rows_f16(SEED_QUERY, 0, T * 3)
    ks = [1, 3, 10, 17, 100]
    want = {k: orc.bruteforce_topk(base, q, k) for k in ks}
    vl = mse.VectorList.from_f16s(base, D)
    disp = mse.Dispatcher(vl, max_wait_us=20000)

    def caller(i):
        res = []
        for rnd in range(3):
            j = rnd * T + i
            k = ks[(i + rnd) % len(ks)]
            res.append((j, k) + disp.search(q[j], k))
        return res

    for res in run_threads(T, caller):
        for j, k, sc, ids in res:
            ws, wi = want[k]
            assert np.array_equal(ids[0], wi[j]) and np.array_equal(sc[0], ws[j])
    st = disp.stats()
    assert st["queries"] == 3 * T and st["requests"] == 3 * T
    assert st["passes"] < 3 * T / 4, st            # callers shared passes (a serial run would need 192)
    assert st["max_pass_queries"] > 8, st          # ... and the matrix-core pass served them
    disp.close()


def test_auto_mode_of_per_thread_searchers_meets_in_the_base_coalescer(gpu, mse, orc):
    """The reference's loop unchanged: a Scratch (searcher) per thread, `mse_bruteforce_topk_f16(.., MSE_MODE_AUTO ..)` with one
    query -- the calls meet in the coalescer the base makes on first use.  A lone caller is answered as well (no company needed)."""
    n, T = 12000, 32
    base = orc.gen_rows_f16(SEED_BASE, 7, n)
    q = orc.gen_rows_f16(SEED_QUERY, 7, T)
    ws, wi = orc.bruteforce_topk(base, q, 10)
    vl = mse.VectorList.from_f16s(base, D)
    searchers = [mse.Searcher(vl) for _ in range(T)]
    sc, ids = searchers[0].bruteforce_topk(q[0], 10)               # alone: fires at once
    assert np.array_equal(ids[0], wi[0]) and np.array_equal(sc[0], ws[0])
    for i, (sc, ids) in enumerate(run_threads(T, lambda i: searchers[i].bruteforce_topk(q[i], 10, mse.MODE_AUTO))):
        assert np.array_equal(ids[0], wi[i]) and np.array_equal(sc[0], ws[i])
    # a multi-query request keeps its rows together
    sc, ids = searchers[1].bruteforce_topk(q[:5], 10, mse.MODE_AUTO)
    assert np.array_equal(ids, wi[:5]) and np.array_equal(sc, ws[:5])


def test_one_callers_error_does_not_leak(gpu, mse, orc):
    """A bad request fails alone (argument errors never enter the queue), and when a SHARED pass fails every request of it is
    repeated on its own: the other callers still get their exact answers."""
    from mse import ffi
    n, T = 5000, 24
    base = orc.gen_rows_f16(SEED_BASE, 3, n)
    q = orc.gen_rows_f16(SEED_QUERY, 3, T)
    ws, wi = orc.bruteforce_topk(base, q, 5)
    vl = mse.VectorList.from_f16s(base, D)
    disp = mse.Dispatcher(vl, max_wait_us=20000)

    def caller(i):
        if i == 5:
            with pytest.raises(mse.MseError, match="k too large"):
                disp.search(q[i], 5000)
            return None
        return disp.search(q[i], 5)

    for i, r in enumerate(run_threads(T, caller)):
        if i != 5:
            assert np.array_equal(r[1][0], wi[i]) and np.array_equal(r[0][0], ws[i])
    # injected failure of the next two shared passes: answers unaffected, the repeats are counted
    ffi.check(ffi.lib().mse_debug_dispatcher_fail_shared(disp._h, 2))
    for i, r in enumerate(run_threads(T, lambda i: disp.search(q[i], 5))):
        assert np.array_equal(r[1][0], wi[i]) and np.array_equal(r[0][0], ws[i])
    assert disp.stats()["retried_alone"] >= 2
    disp.close()


def test_index_searches_from_64_threads_with_an_add_racing_them(gpu, mse, orc):
    """src/main.rs:1016 (`index.write()` in the reload loop) against :1046 (`index.read()` per request): searches share, add
    excludes.  Every search must equal the oracle's answer on SOME prefix of the adds (before or after each one, never a torn
    state), and after the last add on all rows."""
    d, n0, step, n_adds, T, k = 256, 3000, 1000, 4, 64, 7
    rng = np.random.default_rng(11)
    x = (rng.standard_normal((n0 + step * n_adds, d)) / np.sqrt(d)).astype(np.float32)
    q = rng.standard_normal((T, d)).astype(np.float32)
    codes = orc.f16_bits(x)
    prefixes = [n0 + step * a for a in range(n_adds + 1)]
    want = [orc.index_search(codes[:m], q, k, order=0) for m in prefixes]
    idx = mse.ScalarQuantizerIndex(d)
    idx.add(x[:n0])
    stop = threading.Event()

    def caller(i):
        if i == T:                                   # the reload loop
            for a in range(n_adds):
                idx.add(x[n0 + step * a:n0 + step * (a + 1)])
            stop.set()
            return None
        seen = []
        while True:
            done = stop.is_set()
            r = idx.search(q[i], k)
            seen.append((r.distances[0].copy(), r.labels[0].copy()))
            if done:
                return seen

    outs = run_threads(T + 1, caller)
    assert idx.ntotal() == prefixes[-1]
    for i in range(T):
        seen = outs[i]
        last = -1
        for dist, lab in seen:
            hits = [p for p, (wd, wl) in enumerate(want) if np.array_equal(lab, wl[i]) and np.array_equal(dist, wd[i])]
            assert hits, f"caller {i}: a result that matches no prefix of the adds"
            assert max(hits) >= last                  # the index never goes backwards
            last = max(last, min(hits))
        assert len(want) - 1 in [p for p, (wd, wl) in enumerate(want)
                                 if np.array_equal(seen[-1][1], wl[i]) and np.array_equal(seen[-1][0], wd[i])]
    st = idx.stats()
    assert st["passes"] < st["requests"], st          # searches were coalesced


@pytest.mark.parametrize("n,d,nq,k", [(20000, 1152, 40, 10), (5000, 256, 200, 3), (777, 128, 300, 7), (100, 64, 9, 200)])
def test_index_many_queries_take_the_matrix_core_pass_and_match_the_oracle(gpu, mse, orc, n, d, nq, k):
    """nq > 8 on the FAISS surface: one matrix-core pass over f16-rounded queries nominates, the f32 queries re-score in the stated
    order, a certificate (incl. the query rounding) closes it.  Labels AND float distances equal the oracle's; queries are NOT unit
    norm (src/common.rs:215-274)."""
    rng = np.random.default_rng(n + nq)
    x = (rng.standard_normal((n, d)) / np.sqrt(d)).astype(np.float32)
    q = (rng.standard_normal((nq, d)) * rng.uniform(0.2, 5, (nq, 1))).astype(np.float32)
    idx = mse.ScalarQuantizerIndex(d)
    idx.add(x)
    res = idx.search(q, k)
    wd, wl = orc.index_search(orc.f16_bits(x), q, k, order=0)
    assert np.array_equal(res.labels, wl) and np.array_equal(res.distances, wd)
    # the same queries eight at a time (the exact pass) give the same bytes
    for lo in range(0, min(nq, 24), 8):
        r8 = idx.search(q[lo:lo + 8], k)
        assert np.array_equal(r8.labels, wl[lo:lo + 8]) and np.array_equal(r8.distances, wd[lo:lo + 8])


def test_index_query_beyond_f16_range_still_exact(gpu, mse, orc):
    """A query component past the f16 range rounds to infinity in the nomination pass: no certificate, the exact pass answers."""
    rng = np.random.default_rng(5)
    n, d = 4000, 128
    x = (rng.standard_normal((n, d)) / np.sqrt(d)).astype(np.float32)
    q = rng.standard_normal((12, d)).astype(np.float32)
    q[3, 5] = 1.0e6
    q[7] *= 1e-7                                      # and one whose f16 rounding is all subnormals / zeros
    idx = mse.ScalarQuantizerIndex(d)
    idx.add(x)
    res = idx.search(q, 5)
    wd, wl = orc.index_search(orc.f16_bits(x), q, 5, order=0)
    assert np.array_equal(res.labels, wl) and np.array_equal(res.distances, wd)


def test_pq_scan_one_query_per_thread_is_batched_and_unchanged(gpu, mse, orc):
    """mse_pq_scan_topk with one query from 24 threads (own searcher each, two different descriptor-scale vectors, two k):
    the calls meet in the quantiser's coalescer, are grouped by what may share a batch call, and return what the call returns
    when made alone."""
    n, T, r = 30000, 24, 100
    centroids, transform, dpc, d = make_pq(orc)
    rng = np.random.default_rng(9)
    base_f = (rng.standard_normal((n, d)) / np.sqrt(d)).astype(np.float32)
    base = orc.f16_bits(base_f)
    opq, pq = orc.PQ(centroids, transform, dpc, d), mse.ProductQuantizer(centroids, transform, dpc, d)
    codes_h = opq.quantize_batch(orc.f16_to_f32(base))
    desc = rng.integers(0, 256, (n, 4), dtype=np.uint8)
    codes = mse.Codes(codes_h, desc)
    vl = mse.VectorList.from_f16s(base, d)
    q = (rng.standard_normal((T, d)) / np.sqrt(d)).astype(np.float32)
    scales = [np.array([0.5, 0, -0.25, 0.125], np.float32) / np.float32(512), np.array([0, 0.25, 0.125, 0], np.float32) / np.float32(512)]
    searchers = [mse.Searcher(vl) for _ in range(T)]

    def call(i):
        return pq.scan_topk(codes, q[i], r, 10 if i % 3 else 5, searchers[i], scales[i % 2])

    alone = [call(i) for i in range(T)]
    outs = run_threads(T, call)
    for i in range(T):
        assert np.array_equal(outs[i][1], alone[i][1]) and np.array_equal(outs[i][0], alone[i][0])
    # and the answer is the oracle's pipeline: ADC top-r -> exact re-score (+ bias) -> top-k
    for i in (1, 2, 3):
        k = 10 if i % 3 else 5
        sc_i = scales[i % 2]
        approx = opq.adc_desc(opq.preprocess_query(q[i]), codes_h, desc, sc_i)
        _, cand = orc.topk_from_scores(approx, r)
        exact = orc.score_rows(base, cand, orc.f16_bits(q[i])) + np.array([orc.descriptor_product(sc_i, desc, int(c)) for c in cand])
        order = np.lexsort((cand, -exact))[:k]
        assert np.array_equal(outs[i][1], cand[order]) and np.array_equal(outs[i][0], exact[order])


def test_beam_search_one_query_per_thread_is_one_launch_and_unchanged(gpu, mse, orc):
    """The reference's request path: every HTTP request runs ONE greedy_search on its own task (src/query_disk_index.rs:436-540,
    711-736).  32 threads, each with its own searcher, call the batched device search with a single query -- f32 queries (tables
    made on the device) with per-request descriptor scales, and f16 queries with host tables; two different search lists.  The
    calls meet in the graph's coalescer; every caller gets exactly what the same call returns when made alone, which is the
    oracle's greedy_search."""
    from test_gpu_pq_index_graph import clustered_rows, knn_graph, train_pq
    rng = np.random.default_rng(21)
    n, deg, T = 3000, 14, 32
    x = clustered_rows(orc, n, n_centres=32)
    base = orc.f16_bits(x)
    cents, Tm = train_pq(orc, x[:2000], iters=2)
    opq, gpq = orc.PQ(cents, Tm, 18, D), mse.ProductQuantizer(cents, Tm, 18, D)
    codes = opq.quantize_batch(orc.f16_to_f32(base))
    desc = rng.integers(0, 256, size=(n, 4), dtype=np.uint8)
    has_url = (rng.random(n) > 0.1).astype(np.uint8)
    adj, degs = knn_graph(x, deg, rng)
    vl = mse.VectorList.from_f16s(base, D)
    searchers = [mse.Searcher(vl) for _ in range(T)]
    gcodes = mse.Codes(codes, desc)
    dgraph = mse.DeviceGraph(mse.IndexGraph(adj, degs), has_url)
    qs = clustered_rows(orc, T, n_centres=32, seed=201)
    qh = orc.f16_bits(qs)
    starts = rng.integers(0, n, size=T).astype(np.uint32)
    scales = (rng.standard_normal((T, 4)) / 512).astype(np.float32)

    def call(i):
        L = 48 if i % 2 else 64
        if i % 4 < 2:     # f32 query in, per-request scales
            return mse.disk_search_batch(searchers[i], gpq, gcodes, dgraph, starts[i:i + 1], qs[i:i + 1].astype(np.float32), None,
                                         scales[i:i + 1], False, 4, search_list=L, visited_cap=n)[0]
        lut = opq.preprocess_query(qs[i])
        return mse.disk_search_batch(searchers[i], gpq, gcodes, dgraph, starts[i:i + 1], qh[i:i + 1], lut[None], None, False, 2,
                                     search_list=L, visited_cap=n)[0]

    alone = [call(i) for i in range(T)]
    for rnd in range(2):
        outs = run_threads(T, call)
        for i in range(T):
            for a, b in zip(outs[i], alone[i]):
                assert np.array_equal(a, b), (rnd, i)
    i = 2                                                        # f16 + host table: the oracle's search
    obuf, ovids, ovsc, ocm, opc = orc.disk_greedy_search(base, adj, degs, codes, desc, int(starts[i]), qh[i], opq.preprocess_query(qs[i]), None,
                                                          False, 2, 64, has_url)
    bi, bs, vi, vs, cm, pc = alone[i]
    assert (cm, pc) == (ocm, opc) and np.array_equal(bi, obuf.ids) and np.array_equal(bs, obuf.scores) and np.array_equal(vi, ovids)


def test_requests_of_many_sizes_share_the_queue(gpu, mse, orc):
    """Requests larger than a pass (300 queries), mid-sized ones and single queries from 12 threads at once, with a small
    max_queries_per_pass so that passes fill up: a request is never split between callers' rows, one larger than a pass goes alone,
    and every row is the oracle's."""
    n = 9000
    base = orc.gen_rows_f16(SEED_BASE, 11, n)
    q = orc.gen_rows_f16(SEED_QUERY, 11, 700)
    ws, wi = orc.bruteforce_topk(base, q, 6)
    vl = mse.VectorList.from_f16s(base, D)
    disp = mse.Dispatcher(vl, max_queries_per_pass=64, max_wait_us=5000)
    spans = [(0, 300), (300, 301), (301, 340), (340, 341), (341, 400), (400, 401), (401, 402), (402, 470), (470, 471), (471, 600), (600, 601), (601, 700)]

    def caller(i):
        lo, hi = spans[i]
        out = []
        for _ in range(3):
            out.append(disp.search(q[lo:hi], 6))
        return out

    for i, res in enumerate(run_threads(len(spans), caller)):
        lo, hi = spans[i]
        for sc, ids in res:
            assert np.array_equal(ids, wi[lo:hi]) and np.array_equal(sc, ws[lo:hi]), i
    st = disp.stats()
    assert st["queries"] == 3 * 700 and st["requests"] == 3 * len(spans)
    assert st["max_pass_queries"] >= 300            # the 300-query request went through as one pass of its own
    disp.close()


def test_index_near_duplicates_widen_only_their_query(gpu, mse, orc):
    """The FAISS surface with re-posted rows: 700 copies of one row (one per 32-row group) make ONE query of a 40-query batch fail
    its first certificate -- it alone is carried through the wider rounds; a second query with copies in 2200 groups (beyond the
    widest nomination) ends in the exact pass.  Labels (lowest ids among the ties first) and float distances equal the oracle's."""
    rng = np.random.default_rng(17)
    n, d, nq, k = 100_000, 256, 40, 10
    x = (rng.standard_normal((n, d)) / np.sqrt(d)).astype(np.float32)
    q = rng.standard_normal((nq, d)).astype(np.float32)
    groups = rng.permutation(n // 32)
    pos5 = groups[:700] * 32 + rng.integers(0, 32, 700)
    pos9 = groups[700:2900] * 32 + rng.integers(0, 32, 2200)
    x[pos5] = (q[5] / np.linalg.norm(q[5]) * 0.9).astype(np.float32)
    x[pos9] = (q[9] / np.linalg.norm(q[9]) * 0.9).astype(np.float32)
    idx = mse.ScalarQuantizerIndex(d)
    for lo in range(0, n, 8192):
        idx.add(x[lo:lo + 8192])
    res = idx.search(q, k)
    wd, wl = orc.index_search(orc.f16_bits(x), q, k, order=0)
    assert np.array_equal(wl[5], np.sort(pos5)[:k]) and np.array_equal(wl[9], np.sort(pos9)[:k])
    assert np.array_equal(res.labels, wl) and np.array_equal(res.distances, wd)


def _want_topk(ovids, ovsc, k):
    """The server's last step on the oracle's visited list: by (score descending, id ascending), first k, padded."""
    order = sorted(range(len(ovids)), key=lambda j: (-int(ovsc[j]), int(ovids[j])))[:k]
    ids = np.full(k, 0xFFFFFFFF, np.uint32)
    sc = np.full(k, np.iinfo(np.int64).min, np.int64)
    ids[:len(order)] = ovids[order]
    sc[:len(order)] = ovsc[order]
    return ids, sc


def test_request_path_one_query_per_thread_is_shared_and_unchanged(gpu, mse, orc):
    """The metric's own path in the reference's call shape (src/query_disk_index.rs:436-540,711-736; perf_test.py:6-29): 64 request
    threads, ONE query per mse_disk_query_topk(_f32) call, all through ONE searcher handle (a coalesced call only reads its base).
    Mixed k and search lists; f32 queries with per-request descriptor scales, ADC-scored, entry by the reference's shard-centroid rule;
    and f16 queries scored exactly.  Every caller gets the oracle's answer for ITS query -- shard selection, greedy_search, sort --
    and the submissions really are shared."""
    from test_gpu_pq_index_graph import clustered_rows, knn_graph, train_pq
    rng = np.random.default_rng(31)
    n, deg, T, S = 3000, 14, 64, 42
    x = clustered_rows(orc, n, n_centres=32)
    base = orc.f16_bits(x)
    cents, Tm = train_pq(orc, x[:2000], iters=2)
    opq, gpq = orc.PQ(cents, Tm, 18, D), mse.ProductQuantizer(cents, Tm, 18, D)
    codes = opq.quantize_batch(orc.f16_to_f32(base))
    desc = rng.integers(0, 256, size=(n, 4), dtype=np.uint8)
    has_url = (rng.random(n) > 0.1).astype(np.uint8)
    adj, degs = knn_graph(x, deg, rng)
    vl = mse.VectorList.from_f16s(base, D)
    shared = mse.Searcher(vl)
    gcodes = mse.Codes(codes, desc)
    dgraph = mse.DeviceGraph(mse.IndexGraph(adj, degs), has_url)
    # the index header's shards: centroid + medioid (start node) each; two shards share a centroid, so ties exist (last one wins)
    centroids = x[rng.choice(n, S, replace=False)].astype(np.float32) * np.float32(0.9)
    centroids[17] = centroids[5]
    medioids = rng.choice(n, S, replace=False).astype(np.uint32)
    mse.set_entry_centroids(dgraph, centroids, medioids)
    qs = (clustered_rows(orc, T, n_centres=32, seed=301) * np.float32(1.3)).astype(np.float32)
    qs[7] = centroids[5] * np.float32(2.0)                      # a query for which shards 5 and 17 tie exactly
    qh = orc.f16_bits(qs)
    scales = (rng.standard_normal((T, 4)) / 512).astype(np.float32)

    def params(i):
        return (10 if i % 3 else 25), (48 if i % 2 else 64)

    def call(i):
        k, L = params(i)
        if i % 4 < 2:     # the handler as the reference runs it: f32 query, per-request scales, ADC-scored neighbours
            return mse.disk_query_topk(shared, gpq, gcodes, dgraph, qs[i:i + 1], k, None, None, scales[i:i + 1], False, 4, L)
        return mse.disk_query_topk(shared, None, None, dgraph, qh[i:i + 1], k, None, None, None, True, 2, L)

    before = mse.coalescer_stats(dgraph)
    for rnd in range(3):
        outs = run_threads(T, call)
        if rnd:
            continue
        for i in range(T):
            k, L = params(i)
            adc = i % 4 < 2
            # the f16-query form widens the query exactly for the shard selection
            shard = orc.select_shard(centroids, qs[i] if adc else orc.f16_to_f32(qh[i]))
            if i == 7:
                assert shard == 17
            _, ovids, ovsc, ocm, opc = orc.disk_greedy_search(base, adj, degs, codes, desc, int(medioids[shard]), qh[i],
                                                              opq.preprocess_query(qs[i]) if adc else np.zeros(64 * 256, np.float32),
                                                              scales[i] if adc else None, not adc, 4 if adc else 2, L, has_url)
            want_ids, want_sc = _want_topk(ovids, ovsc, k)
            ids, sc, st = outs[i]
            assert np.array_equal(ids[0], want_ids) and np.array_equal(sc[0], want_sc), i
            assert (int(st["cmps"][0]), int(st["pq_cmps"][0]), int(st["n_visited"][0])) == (ocm, opc, len(ovids)), i
    after = mse.coalescer_stats(dgraph)
    requests, passes = after["requests"] - before["requests"], after["passes"] - before["passes"]
    assert requests == 3 * T and after["queries"] - before["queries"] == 3 * T
    assert passes <= requests // 4, (passes, requests)             # 192 one-query requests in a few dozen submissions at most
    assert after["max_pass_queries"] >= 16
    # small multi-query calls share the queue too; a batch call (beyond 16 queries) goes straight to the device: same answers
    lone = call(5)
    assert np.array_equal(lone[0], outs[5][0]) and np.array_equal(lone[1], outs[5][1])
    few = mse.disk_query_topk(shared, None, None, dgraph, qh[2:12], 10, None, None, None, True, 2, 64)
    big = mse.disk_query_topk(mse.Searcher(vl), None, None, dgraph, qh, 10, None, None, None, True, 2, 64)
    assert np.array_equal(few[0], big[0][2:12]) and np.array_equal(few[1], big[1][2:12])
    # one caller's error stays its own
    def bad_or_good(i):
        if i == 3:
            with pytest.raises(mse.MseError):
                mse.disk_query_topk(shared, None, None, dgraph, qh[i:i + 1], 10, None, None, None, True, 9, 64)     # beamwidth 9
            return None
        return mse.disk_query_topk(shared, None, None, dgraph, qh[i:i + 1], 10, None, None, None, True, 2, 64)
    outs2 = run_threads(16, bad_or_good)
    for i in range(16):
        if i != 3:
            assert np.array_equal(outs2[i][0][0], big[0][i]) and np.array_equal(outs2[i][1][0], big[1][i])


def test_entry_table_replaced_under_request_threads(gpu, mse, orc):
    """mse_graph_set_entries / _set_entry_centroids while request threads are calling: every answer is the answer under ONE of the
    tables, never a torn state or a crash (the setters take the entry lock exclusively, calls hold it shared)."""
    from test_gpu_pq_index_graph import clustered_rows, knn_graph
    rng = np.random.default_rng(32)
    n, deg, T = 3000, 12, 12
    x = clustered_rows(orc, n, n_centres=24)
    base = orc.f16_bits(x)
    adj, degs = knn_graph(x, deg, rng)
    vl = mse.VectorList.from_f16s(base, D)
    s = mse.Searcher(vl)
    dgraph = mse.DeviceGraph(mse.IndexGraph(adj, degs))
    tables = [np.sort(rng.choice(n, 30, replace=False)).astype(np.uint32) for _ in range(2)]
    cen = x[rng.choice(n, 9, replace=False)].astype(np.float32)
    med = rng.choice(n, 9, replace=False).astype(np.uint32)
    qh = orc.f16_bits(clustered_rows(orc, T, n_centres=24, seed=302))
    want = []
    for t in tables:
        mse.set_entries(dgraph, vl, t)
        want.append(mse.disk_query_topk(mse.Searcher(vl), None, None, dgraph, qh, 10, None, None, None, True, 2, 32)[0])
    mse.set_entry_centroids(dgraph, cen, med)
    want.append(mse.disk_query_topk(mse.Searcher(vl), None, None, dgraph, qh, 10, None, None, None, True, 2, 32)[0])
    stop = threading.Event()
    bad = []

    def caller(i):
        while not stop.is_set():
            ids = mse.disk_query_topk(s, None, None, dgraph, qh[i:i + 1], 10, None, None, None, True, 2, 32)[0][0]
            if not any(np.array_equal(ids, w[i]) for w in want):
                bad.append(i)

    ts = [threading.Thread(target=caller, args=(i,)) for i in range(T)]
    for t in ts:
        t.start()
    for r in range(12):
        if r % 3 == 2:
            mse.set_entry_centroids(dgraph, cen, med)
        else:
            mse.set_entries(dgraph, vl, tables[r % 3])
    stop.set()
    for t in ts:
        t.join()
    assert not bad, bad


def test_request_path_without_a_thread_per_request(gpu, mse, orc):
    """mse_disk_query_submit_f32 / mse_graph_completions (mse.QueryTickets): ONE host thread keeps hundreds of one-query requests in
    flight, the way the reference's monoio tasks would (src/query_disk_index.rs:640-655,716-732).  Every ticket comes back exactly
    once with the answer the batch call gives for ITS query (ADC-scored with per-request scales, and exactly scored), the requests
    share submissions, blocking callers on other threads are served beside them, a refused request fails alone with its own
    message, and a poll with nothing in flight returns at once."""
    from test_gpu_pq_index_graph import clustered_rows, knn_graph, train_pq
    rng = np.random.default_rng(41)
    n, deg, Q = 4000, 14, 600
    x = clustered_rows(orc, n, n_centres=32)
    base = orc.f16_bits(x)
    cents, Tm = train_pq(orc, x[:2000], iters=2)
    gpq = mse.ProductQuantizer(cents, Tm, 18, D)
    codes = orc.PQ(cents, Tm, 18, D).quantize_batch(orc.f16_to_f32(base))
    desc = rng.integers(0, 256, size=(n, 4), dtype=np.uint8)
    adj, degs = knn_graph(x, deg, rng)
    vl = mse.VectorList.from_f16s(base, D)
    s = mse.Searcher(vl)
    gcodes = mse.Codes(codes, desc)
    dgraph = mse.DeviceGraph(mse.IndexGraph(adj, degs), None)
    mse.set_entries(dgraph, vl, np.sort(rng.choice(n, 64, replace=False)).astype(np.uint32))
    qs = (clustered_rows(orc, Q, n_centres=32, seed=401) * np.float32(1.2)).astype(np.float32)
    scales = (rng.standard_normal((Q, 4)) / 512).astype(np.float32)
    want_adc = mse.disk_query_topk(mse.Searcher(vl), gpq, gcodes, dgraph, qs, 10, None, None, scales, False, 4, 48)
    want_exact = mse.disk_query_topk(mse.Searcher(vl), None, None, dgraph, qs, 7, None, None, None, True, 2, 32)
    adc = mse.QueryTickets(s, gpq, gcodes, dgraph, 10, False, 4, 48)
    exact = mse.QueryTickets(s, None, None, dgraph, 7, True, 2, 32)
    assert adc.collect(timeout_us=0) == []                       # nothing in flight: a poll returns at once
    # the completion descriptor: quiet now, readable once a request has completed, quiet again after its counter is read
    import os
    import select
    fd = adc.fileno()
    assert fd >= 0 and exact.fileno() == fd and select.select([fd], [], [], 0)[0] == []
    exact.submit(qs[0], key="probe")
    assert select.select([fd], [], [], 5.0)[0] == [fd]
    assert int.from_bytes(os.read(fd, 8), "little") >= 1
    got_probe = exact.collect(timeout_us=0)
    assert [k_ for k_, _, _ in got_probe] == ["probe"] and np.array_equal(got_probe[0][1][0], want_exact[0][0])
    assert select.select([fd], [], [], 0)[0] == [] and exact.collect(timeout_us=0) == []
    before = mse.coalescer_stats(dgraph)
    # blocking callers on other threads share the queue with the tickets
    stop, blocked_bad = threading.Event(), []

    def blocking(i):
        while not stop.is_set():
            ids = mse.disk_query_topk(s, None, None, dgraph, qs[i:i + 1], 7, None, None, None, True, 2, 32)[0]
            if not np.array_equal(ids[0], want_exact[0][i]):
                blocked_bad.append(i)

    ts = [threading.Thread(target=blocking, args=(i,)) for i in range(8)]
    for t in ts:
        t.start()
    got = {}
    window, nxt = 256, 0
    while len(got) < 2 * Q:
        while adc.in_flight < window and nxt < 2 * Q:
            i, kind = nxt // 2, nxt % 2
            if kind == 0:
                adc.submit(qs[i], scales[i], key=("adc", i))
            else:
                exact.submit(qs[i], key=("exact", i))
            nxt += 1
        # a graph has ONE completion list: either object's collect() returns whatever has completed, of both kinds
        buf = (adc if nxt % 3 else exact).collect(max_tickets=64, timeout_us=2_000_000)
        assert buf, "no completion within two seconds"
        for key, ids, sc in buf:
            assert key not in got
            got[key] = (ids, sc)
    stop.set()
    for t in ts:
        t.join()
    assert not blocked_bad
    for i in range(Q):
        assert np.array_equal(got[("adc", i)][0][0], want_adc[0][i]) and np.array_equal(got[("adc", i)][1][0], want_adc[1][i]), i
        assert np.array_equal(got[("exact", i)][0][0], want_exact[0][i]) and np.array_equal(got[("exact", i)][1][0], want_exact[1][i]), i
    after = mse.coalescer_stats(dgraph)
    assert after["requests"] - before["requests"] >= 2 * Q
    assert after["passes"] - before["passes"] <= (after["requests"] - before["requests"]) // 8      # shared submissions
    # completion queues of their own (one per event loop of a host that runs several): each object gets back exactly what IT submitted,
    # through its own descriptor, while the graph's shared list stays empty
    loop_a = mse.QueryTickets(s, None, None, dgraph, 7, True, 2, 32, own_queue=True)
    loop_b = mse.QueryTickets(s, gpq, gcodes, dgraph, 10, False, 4, 48, own_queue=True)
    assert loop_a.fileno() != loop_b.fileno() and loop_a.fileno() != exact.fileno()
    for i in range(40):
        loop_a.submit(qs[i], key=("a", i), copy=bool(i % 2))                # every other one read where it lies (_nocopy)
        loop_b.submit(qs[i], scales[i], key=("b", i))
    got_a, got_b = {}, {}
    while len(got_a) < 40:
        for key, ids, sc in loop_a.collect(timeout_us=2_000_000):
            got_a[key] = ids
    while len(got_b) < 40:
        for key, ids, sc in loop_b.collect(timeout_us=2_000_000):
            got_b[key] = ids
    assert set(got_a) == {("a", i) for i in range(40)} and set(got_b) == {("b", i) for i in range(40)}
    assert all(np.array_equal(got_a[("a", i)][0], want_exact[0][i]) and np.array_equal(got_b[("b", i)][0], want_adc[0][i]) for i in range(40))
    assert exact.collect(timeout_us=0) == [] and loop_a.collect(timeout_us=0) == []
    loop_a.close()
    loop_b.close()
    # a refused request (search list beyond the limit) is refused at submit; one that fails when executed fails alone
    with pytest.raises(mse.MseError):
        mse.QueryTickets(s, None, None, dgraph, 7, True, 2, 5000).submit(qs[0])
    with pytest.raises(mse.MseError):
        mse.QueryTickets(s, None, None, dgraph, 7, True, 2, 32).submit(qs[:17])                     # more than 16 queries per request
    assert adc.in_flight == 0 and exact.in_flight == 0
