"""Parity of the device Vamana build (diskann/src/lib.rs:183-389) with the CPU oracle: bit-exact graphs."""
import numpy as np
import pytest

from conftest import SEED_BASE, SEED_QUERY
from test_gpu_pq_index_graph import clustered_rows

pytestmark = pytest.mark.gpu
D = 1152


def rows(orc, n, clustered=True, seed=0):
    if not clustered:
        return orc.gen_rows_f16(SEED_BASE, 0, n)
    return orc.f16_bits(clustered_rows(orc, n, n_centres=max(8, n // 60), noise=0.5, seed=seed))


def cfg_pair(orc, mse, **kw):
    return orc.BuildConfig.make(**kw), mse.IndexBuildConfig(**kw)


def test_random_fill_matches_oracle(gpu, mse, orc):
    n, r = 5000, 64
    g = mse.BuildGraph(n, r)
    g.random_fill(0xABC)
    h = g.to_host()
    adj, deg = orc.random_fill_graph(0xABC, n, r)
    assert np.array_equal(h.deg, deg) and np.array_equal(h.adj, adj)
    assert (deg == r).all() and all(len(set(row)) == r for row in adj[:200])
    # a partly filled list is topped up, not rebuilt
    adj2, deg2 = np.zeros((n, r), np.uint32), np.zeros(n, np.uint32)
    adj2[:, 0] = 7
    deg2[:] = 1
    g2 = mse.BuildGraph(n, r, mse.IndexGraph(adj2, deg2))
    g2.random_fill(5)
    h2 = g2.to_host()
    a3, d3 = orc.random_fill_graph(5, n, r, adj2.copy(), deg2.copy())
    assert (h2.deg == r).all() and np.array_equal(h2.adj, a3) and (h2.adj[:, 0] == 7).all()


@pytest.mark.parametrize("base_only", [False, True])
def test_graph_search_batch_matches_oracle(gpu, mse, orc, base_only):
    n, r, L, nq = 4000, 24, 75, 40
    vecs = rows(orc, n)
    adj, deg = orc.random_fill_graph(11, n, r)
    adj[5, 3] = adj[5, 1]   # an id listed twice in one list
    deg[17] = 5             # a short list
    qb = 3600 if base_only else 0xFFFFFFFF
    queries = orc.gen_rows_f16(SEED_QUERY, 0, nq)
    queries[:8] = vecs[100:108]
    s = mse.Searcher(mse.VectorList.from_f16s(vecs, D))
    g = mse.BuildGraph(n, r, mse.IndexGraph(adj, deg))
    starts = np.arange(nq, dtype=np.uint32) * 7
    got = g.search_batch(s, starts, queries, L, base_vectors_only=base_only, query_breakpoint=qb)
    for i in range(nq):
        nb, dist = orc.greedy_search(vecs, adj, deg, int(starts[i]), queries[i], L, base_vectors_only=base_only, query_breakpoint=qb)
        assert np.array_equal(got[i][0], nb.ids) and np.array_equal(got[i][1], nb.scores), i
        assert got[i][2] == dist
        if base_only:
            assert (got[i][0][got[i][0] != starts[i]] < qb).all()


@pytest.mark.parametrize("n_cand,alpha,saturate,p_is_query", [(300, 65536, False, False), (5000, 78643, False, False),
                                                               (2500, 65536, True, False), (900, 70000, False, True), (0, 65536, False, False)])
def test_robust_prune_matches_oracle(gpu, mse, orc, n_cand, alpha, saturate, p_is_query):
    n = 3000
    vecs = rows(orc, n, seed=3)
    s = mse.Searcher(mse.VectorList.from_f16s(vecs, D))
    rng = np.random.default_rng(n_cand + alpha)
    p = 2900 if p_is_query else 77
    kw = dict(r=64, l=192, maxc=750, alpha=alpha, query_alpha=90000, saturate_graph=saturate, query_breakpoint=2800)
    oc, mc = cfg_pair(orc, mse, **kw)
    ids = rng.integers(0, n, n_cand).astype(np.uint32)   # duplicates and p itself occur, as in a real visited list
    if n_cand:
        ids[5] = p
        ids[40] = ids[41]
    scores = orc.score_rows(vecs, ids, vecs[p]) if n_cand else np.empty(0, np.int64)
    want = orc.robust_prune(vecs, ids, scores, p, oc)
    got = mse.robust_prune(s, ids, scores, p, mc)
    assert np.array_equal(got, want)
    if n_cand:
        assert 0 < len(want) <= 64


def test_robust_prune_mfma_route_long_list(gpu, mse, orc, monkeypatch):
    """The candidate-major MFMA walk behind a 6000-entry list (value cut + one sort), default bound and widened bound."""
    n = 3000
    vecs = rows(orc, n, seed=4)
    s = mse.Searcher(mse.VectorList.from_f16s(vecs, D))
    rng = np.random.default_rng(8)
    ids = rng.integers(0, n, 6000).astype(np.uint32)
    scores = orc.score_rows(vecs, ids, vecs[11])
    kw = dict(r=64, l=192, maxc=750, alpha=65536, query_alpha=65536, query_breakpoint=0xFFFFFFFF)
    oc, mc = cfg_pair(orc, mse, **kw)
    want = orc.robust_prune(vecs, ids, scores, 11, oc)
    assert np.array_equal(mse.robust_prune(s, ids, scores, 11, mc), want)
    monkeypatch.setenv("MSE_GRAM_EPS_SCALE", "3000")
    assert np.array_equal(mse.robust_prune(s, ids, scores, 11, mc), want)
    monkeypatch.setenv("MSE_BUILD_EXACT_PRUNE", "1")
    assert np.array_equal(mse.robust_prune(s, ids, scores, 11, mc), want)


def test_certificate_covers_subnormal_components(gpu, mse, orc):
    """A decision that hinges on 1151 products of a normal value with an f16 SUBNORMAL one (6e-5 < 2^-14).  If the matrix
    cores flushed subnormal inputs, their sum would miss 3.45e-3 here -- more than the stated error bound -- and the MFMA route
    would keep a candidate the reference discards.  The exact walk is the referee."""
    s_row = np.full(D, 0.05, np.float32); s_row[0] = 0.9          # the first p_star
    dummy = np.zeros(D, np.float32); dummy[0] = 0.5               # sits right behind it (never tested against it)
    c_row = np.full(D, 6e-5, np.float32); c_row[0] = 0.02         # dot with s_row: 0.018 + 1151 * 0.05 * 6e-5 = 0.02145 >= 0.02
    p_row = np.zeros(D, np.float32); p_row[0] = 1.0
    filler = np.zeros((60, D), np.float32); filler[:, 1] = 1.0    # keeps the base from being tiny
    vecs = orc.f16_bits(np.stack([p_row, s_row, dummy, c_row, *filler]))
    assert (orc.f16_to_f32(vecs[3][1:]) > 0).all() and (orc.f16_to_f32(vecs[3][1:]) < 2.0 ** -14).all()   # really subnormal
    s = mse.Searcher(mse.VectorList.from_f16s(vecs, D))
    ids = np.array([1, 2, 3], np.uint32)
    scores = orc.score_rows(vecs, ids, vecs[0])
    oc, mc = cfg_pair(orc, mse, r=8, l=16, maxc=16)
    want = orc.robust_prune(vecs, ids, scores, 0, oc)
    assert want.tolist() == [1, 2]                                 # the reference discards candidate 3
    assert mse.robust_prune(s, ids, scores, 0, mc).tolist() == [1, 2]


def build_both(orc, mse, vecs, r, order, med, passes, batch, seed=21, stitch_order=None):
    n = len(vecs)
    adj, deg = orc.random_fill_graph(seed, n, r)
    s = mse.Searcher(mse.VectorList.from_f16s(vecs, D))
    g = mse.BuildGraph(n, r)
    g.random_fill(seed)
    for kw in passes:
        oc, mc = cfg_pair(orc, mse, **kw)
        orc.build_graph(vecs, adj, deg, order, med, oc, batch)
        g.build(s, order, med, mc, batch)
    if stitch_order is not None:
        orc.robust_stitch(vecs, adj, deg, stitch_order, oc)
        g.robust_stitch(s, stitch_order, mc)
    return adj, deg, g, s


def test_build_graph_sequential_matches_oracle(gpu, mse, orc):
    """batch = 1: the reference's single-threaded loop (lib.rs:294,297), graph equal edge for edge."""
    n, r = 700, 16
    vecs = rows(orc, n, seed=5)
    order = np.random.default_rng(1).permutation(n).astype(np.uint32)
    med = int(orc.medioid(vecs))
    adj, deg, g, _ = build_both(orc, mse, vecs, r, order, med, [dict(r=r, l=40, maxc=90)], 1)
    h = g.to_host()
    assert np.array_equal(h.deg, deg)
    for i in range(n):
        assert np.array_equal(h.adj[i, :deg[i]], adj[i, :deg[i]]), i
    assert deg.max() <= r and deg.min() > 0


def test_build_graph_batched_two_passes_matches_oracle(gpu, mse, orc):
    """Batched form, two passes (alpha 1.0 then 1.2, generate_index_shard.rs:113-127): equal to the oracle's batched form."""
    n, r = 5000, 32
    vecs = rows(orc, n, seed=6)
    order = np.random.default_rng(2).permutation(n).astype(np.uint32)
    med = int(orc.medioid(vecs))
    passes = [dict(r=r, l=64, maxc=300), dict(r=r, l=64, maxc=300, alpha=78643)]
    adj, deg, g, s = build_both(orc, mse, vecs, r, order, med, passes, 128)
    h = g.to_host()
    assert np.array_equal(h.deg, deg)
    for i in range(n):
        assert np.array_equal(h.adj[i, :deg[i]], adj[i, :deg[i]]), i
    # the graph is navigable: recall@1 of self-queries (the reference's harness, diskann/src/main.rs:117-136)
    qi = np.arange(0, n, 25)
    res = g.search_batch(s, med, vecs[qi], 64)
    assert np.mean([res[k][0][0] == qi[k] for k in range(len(qi))]) > 0.85


def test_build_graph_with_queries_and_stitch_matches_oracle(gpu, mse, orc):
    """OOD-DiskANN variant: query nodes behind query_breakpoint, saturated lists for them, then robust_stitch."""
    nb_, nq_, r = 1800, 200, 16
    n = nb_ + nq_
    vecs = np.concatenate([rows(orc, nb_, seed=8), orc.gen_rows_f16(SEED_QUERY, 0, nq_)])
    order = np.random.default_rng(3).permutation(n).astype(np.uint32)
    qorder = (nb_ + np.random.default_rng(4).permutation(nq_)).astype(np.uint32)
    med = int(orc.medioid(vecs[:nb_]))
    kw = dict(r=r, l=48, maxc=120, query_alpha=70000, query_breakpoint=nb_, max_add_per_stitch_iter=3)
    adj, deg, g, _ = build_both(orc, mse, vecs, r, order, med, [kw], 64, stitch_order=qorder)
    h = g.to_host()
    assert np.array_equal(h.deg, deg)
    for i in range(n):
        assert np.array_equal(h.adj[i, :deg[i]], adj[i, :deg[i]]), i
    assert deg.max() <= r


@pytest.mark.parametrize("env", [{"MSE_GRAM_EPS_SCALE": "3000"}, {"MSE_BUILD_EXACT_BACKEDGE": "1"}, {"MSE_BUILD_EXACT_PRUNE": "1"},
                                 {"MSE_BUILD_EXACT_PRUNE": "1", "MSE_BUILD_EXACT_BACKEDGE": "1"}])
def test_mfma_and_exact_routes_agree(gpu, mse, orc, env, monkeypatch):
    """Both prunes (of an inserted point's candidates, of a full list plus a newcomer) take candidate products from MFMA tiles
    and settle comparisons inside the error bound with the exact dot.  Widening the bound 3000-fold sends nearly every
    comparison down the exact path; the all-exact kernels are further routes.  Every route must give the oracle's graph
    (the default routes are what the other tests run)."""
    n, r = 3000, 64
    vecs = rows(orc, n, seed=9)
    order = np.random.default_rng(5).permutation(n).astype(np.uint32)
    med = int(orc.medioid(vecs))
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    adj, deg, g, _ = build_both(orc, mse, vecs, r, order, med, [dict(r=r, l=96, maxc=400)], 256)
    h = g.to_host()
    assert np.array_equal(h.deg, deg)
    for i in range(n):
        assert np.array_equal(h.adj[i, :deg[i]], adj[i, :deg[i]]), i


@pytest.mark.parametrize("d,r,n,batch", [(128, 8, 600, 32), (64, 64, 300, 16), (256, 5, 40, 64)])
def test_build_graph_other_shapes(gpu, mse, orc, d, r, n, batch):
    """Widths other than 1152, degree bounds that are not multiples of 16, a batch larger than the point count, maxc below
    the list length of the back-edge prune: the generic paths, against the oracle."""
    rng = np.random.default_rng(d + r)
    centres = rng.standard_normal((6, d))
    x = centres[rng.integers(0, 6, n)] + 0.7 * rng.standard_normal((n, d))
    vecs = orc.f16_bits((x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32))
    order = rng.permutation(n).astype(np.uint32)
    med = int(orc.medioid(vecs))
    adj, deg = orc.random_fill_graph(2, n, r)
    s = mse.Searcher(mse.VectorList.from_f16s(vecs, d))
    g = mse.BuildGraph(n, r)
    g.random_fill(2)
    for maxc in (40, max(2, r // 2)):
        kw = dict(r=r, l=24, maxc=maxc, alpha=70000)
        orc.build_graph(vecs, adj, deg, order, med, orc.BuildConfig.make(**kw), batch)
        g.build(s, order, med, mse.IndexBuildConfig(**kw), batch)
        h = g.to_host()
        assert np.array_equal(h.deg, deg)
        for i in range(n):
            assert np.array_equal(h.adj[i, :deg[i]], adj[i, :deg[i]]), (maxc, i)


def test_build_properties_at_scale(gpu, mse, orc):
    """1e5 points at the reference's defaults (R 64, L 192, C 750) -- too many for the oracle in a test, so size-independent
    properties instead: the build is a function of (order, initial graph) -- two runs give the same graph, and a different
    batch size a different but equally valid one; lists hold at most R ids, all inside the index; every point finds itself."""
    n, r = 100_000, 64
    host = rows(orc, n, seed=13)
    vecs = mse.VectorList.from_f16s(host, D)
    s = mse.Searcher(vecs)
    med = mse.medioid(vecs)
    order = np.random.default_rng(11).permutation(n).astype(np.uint32)
    cfg = mse.IndexBuildConfig(r=r, l=96, maxc=400)

    def run(batch):
        g = mse.BuildGraph(n, r)
        g.random_fill(3)
        g.build(s, order, med, cfg, batch)
        return g, g.to_host()

    g1, h1 = run(2048)
    _, h2 = run(2048)
    assert np.array_equal(h1.deg, h2.deg) and np.array_equal(h1.adj, h2.adj)
    _, h3 = run(1024)
    assert not np.array_equal(h1.adj, h3.adj)
    for h in (h1, h3):
        assert h.deg.max() <= r and h.deg.min() >= 1
        mask = np.arange(r)[None, :] < h.deg[:, None]
        assert (h.adj[mask] < n).all()
    qi = np.arange(0, n, 200)
    ids, _, _, _ = g1.search_batch(s, med, host[qi], 96, as_arrays=True)
    assert (ids[:, 0] == qi).mean() > 0.97


def test_build_graph_rejects_bad_arguments(gpu, mse, orc):
    n, r = 300, 8
    vecs = rows(orc, n, clustered=False)
    s = mse.Searcher(mse.VectorList.from_f16s(vecs, D))
    g = mse.BuildGraph(n, r)
    g.random_fill(1)
    with pytest.raises(mse.MseError):
        g.build(s, np.arange(n, dtype=np.uint32), 0, mse.IndexBuildConfig(r=16, l=32, maxc=50))      # r != stride
    with pytest.raises(mse.MseError):
        g.build(s, np.array([n], np.uint32), 0, mse.IndexBuildConfig(r=r, l=32, maxc=50))            # point out of range
    with pytest.raises(mse.MseError):
        g.build(s, np.arange(n, dtype=np.uint32), 0, mse.IndexBuildConfig(r=r, l=4096, maxc=50))     # list too long
    bad = np.zeros((n, r), np.uint32)
    bad[3, 0] = n + 5
    g2 = mse.BuildGraph(n, r, mse.IndexGraph(bad, np.full(n, r, np.uint32)))
    with pytest.raises(mse.MseError):
        g2.build(s, np.arange(n, dtype=np.uint32), 0, mse.IndexBuildConfig(r=r, l=32, maxc=50))


def test_back_edges_grouped_on_the_device_hubs_and_ragged_batches(gpu, mse, orc):
    """The back edges of a batch are grouped by target on the device (round 5: an open-addressing table, atomics, segments sorted by
    entry number).  One batch that holds EVERY point of a small set sends hundreds of back edges to the same few lists (every search
    starts at the medioid of a random graph) -- the workgroup-per-target ordering kernel -- next to thousands of targets with one or two;
    a second run uses a batch size that does not divide the point count.  Edge for edge against the oracle's batched form."""
    n, r = 900, 16
    vecs = rows(orc, n, seed=12)
    order = np.random.default_rng(7).permutation(n).astype(np.uint32)
    med = int(orc.medioid(vecs))
    for batch in (n, 257):
        adj, deg, g, _ = build_both(orc, mse, vecs, r, order, med, [dict(r=r, l=40, maxc=90)], batch, seed=22)
        h = g.to_host()
        assert np.array_equal(h.deg, deg), batch
        for i in range(n):
            assert np.array_equal(h.adj[i, :deg[i]], adj[i, :deg[i]]), (batch, i)
