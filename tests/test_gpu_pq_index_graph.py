"""Parity of the HIP product-quantiser, flat index and graph-search paths with the CPU oracle."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, SEED_BASE, SEED_QUERY, SEED_CENTRES, make_pq

pytestmark = pytest.mark.gpu
D = 1152


def clustered_rows(orc, n, n_centres=64, noise=0.3, seed=0):
    """Unit rows around `n_centres` unit centres (iid Gaussian data is not quantisable)."""
    rng = np.random.default_rng(seed)
    centres = orc.f16_to_f32(orc.gen_rows_f16(SEED_CENTRES, 0, n_centres))
    x = centres[rng.integers(0, n_centres, n)] + rng.standard_normal((n, D)).astype(np.float32) * np.float32(noise / np.sqrt(D))
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return x.astype(np.float32)


def train_pq(orc, sample, dpc=18, n_centroids=256, iters=4, seed=1):
    """Tiny OPQ-shaped codec: random orthonormal transform, per-subspace max-inner-product k-means."""
    rng = np.random.default_rng(seed)
    d = sample.shape[1]
    T = np.linalg.qr(rng.standard_normal((d, d)))[0].astype(np.float32)
    t = sample @ T.T                                   # rows = T x
    n_chunks = d // dpc
    cents = np.zeros((n_centroids, d), np.float32)
    for i in range(n_chunks):
        sub = t[:, i * dpc:(i + 1) * dpc]
        c = sub[rng.choice(len(sub), n_centroids, replace=False)].copy()
        for _ in range(iters):
            a = np.argmax(sub @ c.T, axis=1)
            for j in range(n_centroids):
                m = sub[a == j]
                if len(m):
                    c[j] = m.mean(axis=0)
        cents[:, i * dpc:(i + 1) * dpc] = c
    return cents, T


def test_golden_adc(gpu, mse):
    g = np.load(os.path.join(GOLDEN, "pq_adc_4096.npz"))
    pq = mse.ProductQuantizer(np.zeros((256, D), np.float32), np.eye(D, dtype=np.float32), 18, D)
    got = pq.asymmetric_dot_product(g["lut"], g["codes"])
    assert np.array_equal(got, g["adc"])
    codes = mse.Codes(g["codes"], g["desc"])
    ids = np.arange(4096, dtype=np.uint32)[::-1].copy()
    assert np.array_equal(pq.adc_gather(codes, g["lut"], ids, g["scales"]), g["adc_desc"][::-1])
    assert np.array_equal(pq.adc_gather(codes, g["lut"], ids[:100]), g["adc"][::-1][:100])


@pytest.mark.parametrize("d,dpc,nc", [(1152, 18, 256), (128, 16, 8), (64, 16, 4)])
def test_codec_matches_oracle(gpu, mse, orc, d, dpc, nc):
    cents, T, _, _ = make_pq(orc, d, dpc, nc)
    rng = np.random.default_rng(3)
    x = (rng.standard_normal((70, d)) / np.sqrt(d)).astype(np.float32)
    opq, gpq = orc.PQ(cents, T, dpc, d), mse.ProductQuantizer(cents, T, dpc, d)
    assert np.array_equal(gpq.apply_transform(x), opq.apply_transform(x))          # same k-ascending fma order
    assert np.array_equal(gpq.quantize_batch(x), opq.quantize_batch(x))
    lut_g, lut_o = gpq.preprocess_query(x[0]).table, opq.preprocess_query(x[0])
    assert np.array_equal(lut_g, lut_o)
    codes = opq.quantize_batch(x)
    assert np.array_equal(gpq.asymmetric_dot_product(lut_g, codes), opq.asymmetric_dot_product(lut_o, codes))


def test_quantize_tie_rule_and_errors(gpu, mse, orc):
    d, dpc = 64, 16
    T = np.eye(d, dtype=np.float32)
    cents = np.zeros((4, d), np.float32)
    cents[1] = 1.0
    cents[2] = 1.0
    cents[3] = -1.0
    pq = mse.ProductQuantizer(cents, T, dpc, d)
    assert list(pq.quantize_batch(np.ones((1, d), np.float32))[0]) == [1] * 4     # first max wins (vector.rs:353-358)
    assert list(pq.quantize_batch(np.zeros((1, d), np.float32))[0]) == [0] * 4
    assert list(pq.quantize_batch(-np.ones((1, d), np.float32))[0]) == [3] * 4


def test_pq_scan_rerank_recall(gpu, mse, orc):
    """BASELINE config 5 at test size: ADC scan -> top-r -> fp16 exact re-score -> top-10; recall@10
    against the oracle's brute force, and bit parity of every stage with the oracle pipeline."""
    n, r, k = 20000, 200, 10
    x = clustered_rows(orc, n)
    cents, T = train_pq(orc, x[:4000])
    base = orc.f16_bits(x)
    opq, gpq = orc.PQ(cents, T, 18, D), mse.ProductQuantizer(cents, T, 18, D)
    codes = opq.quantize_batch(orc.f16_to_f32(base))
    assert np.array_equal(gpq.quantize_batch(orc.f16_to_f32(base[:512])), codes[:512])
    desc = np.random.default_rng(2).integers(0, 256, size=(n, 4), dtype=np.uint8)
    scales = np.array([0.5, 0, -0.25, 0], np.float32) / np.float32(512)
    gcodes = mse.Codes(codes, desc)
    searcher = mse.Searcher(mse.VectorList.from_f16s(base, D))
    queries = clustered_rows(orc, 12, seed=9)
    hits = 0
    for qv in queries:
        qh = orc.f16_bits(qv)
        # oracle pipeline
        lut = opq.preprocess_query(qv)
        approx = opq.adc_desc(lut, codes, desc, scales)
        _, cand = orc.topk_from_scores(approx, r)
        exact = orc.score_rows(base, cand, qh) + np.array([orc.descriptor_product(scales, desc, int(c)) for c in cand])
        order = np.lexsort((cand, -exact))[:k]
        sc, ids = gpq.scan_topk(gcodes, qv, r, k, searcher, scales)
        assert np.array_equal(ids, cand[order]) and np.array_equal(sc, exact[order])
        # without re-score: plain ADC ranking
        sc2, ids2 = gpq.scan_topk(gcodes, qv, r, k, None, scales)
        ws, wi = orc.topk_from_scores(approx, k)
        assert np.array_equal(ids2, wi) and np.array_equal(sc2, ws)
        # recall against exact brute force on the same scoring (dot + descriptor bias)
        truth = orc.score_all(base, qh) + np.array([orc.descriptor_product(scales, desc, i) for i in range(n)])
        ranks = orc.ranks_from_scores(truth)
        hits += int(np.sum(ranks[ids] < k))
    recall = hits / (k * len(queries))
    assert recall >= 0.9, recall


def test_pq_scan_batch_with_exact_rescore_equals_one_by_one(gpu, mse, orc):
    """mse_pq_scan_topk_batch alternates its queries between two streams with separate scratch (the tail of one query runs beside
    the next query's scan): every query's answer -- ADC top-r, exact fp16 re-score with descriptor bias, top-k -- must equal the
    one-query-per-call answer and the oracle pipeline, in both the re-scored and the ADC-only form, for odd and even batch sizes."""
    n, r, k = 9000, 120, 10
    x = clustered_rows(orc, n, n_centres=32)
    cents, T = train_pq(orc, x[:3000], iters=2)
    base = orc.f16_bits(x)
    opq, gpq = orc.PQ(cents, T, 18, D), mse.ProductQuantizer(cents, T, 18, D)
    codes = opq.quantize_batch(orc.f16_to_f32(base))
    desc = np.random.default_rng(3).integers(0, 256, size=(n, 4), dtype=np.uint8)
    scales = np.array([0.5, 0, -0.25, 0.125], np.float32) / np.float32(512)
    gcodes = mse.Codes(codes, desc)
    searcher = mse.Searcher(mse.VectorList.from_f16s(base, D))
    queries = clustered_rows(orc, 9, n_centres=32, seed=11)
    for nq in (9, 4, 3):
        bs, bi = gpq.scan_topk_batch(gcodes, queries[:nq], r, k, searcher, scales)
        as_, ai = gpq.scan_topk_batch(gcodes, queries[:nq], r, k, None, scales)
        for j in range(nq):
            qv = queries[j]
            s1, i1 = gpq.scan_topk(gcodes, qv, r, k, searcher, scales)
            assert np.array_equal(bs[j], s1) and np.array_equal(bi[j], i1), (nq, j)
            approx = opq.adc_desc(opq.preprocess_query(qv), codes, desc, scales)
            _, cand = orc.topk_from_scores(approx, r)
            exact = orc.score_rows(base, cand, orc.f16_bits(qv)) + np.array([orc.descriptor_product(scales, desc, int(c)) for c in cand])
            order = np.lexsort((cand, -exact))[:k]
            assert np.array_equal(bi[j], cand[order]) and np.array_equal(bs[j], exact[order]), (nq, j)
            ws, wi = orc.topk_from_scores(approx, k)
            assert np.array_equal(ai[j], wi) and np.array_equal(as_[j], ws), (nq, j)


@pytest.mark.parametrize("n,r,k", [(70, 200, 10), (4097, 64, 64), (12345, 200, 10), (64, 5, 5), (1, 3, 2)])
def test_pq_scan_group_maxima_ragged_ties_and_batch(gpu, mse, orc, n, r, k):
    """The flat scan keeps one maximum per 64 vectors and re-scores the best groups: ragged sizes (fewer groups than r, a partial
    last group, fewer vectors than k), heavy score ties across groups (few distinct code rows) and the batched entry point,
    all against the oracle's ADC scores ranked by (score desc, id asc)."""
    rng = np.random.default_rng(n)
    cents, T, _, _ = make_pq(orc)
    opq, gpq = orc.PQ(cents, T, 18, D), mse.ProductQuantizer(cents, T, 18, D)
    distinct = rng.integers(0, 256, size=(7, 64), dtype=np.uint8)
    codes = distinct[rng.integers(0, 7, size=n)]                      # many vectors share a code row => equal ADC scores
    desc = rng.integers(0, 3, size=(n, 4), dtype=np.uint8)
    scales = np.array([0.25, 0, -0.125, 0.5], np.float32) / np.float32(512)
    gcodes = mse.Codes(codes, desc)
    qs = (rng.standard_normal((5, D)) / np.sqrt(D)).astype(np.float32)
    bs, bi = gpq.scan_topk_batch(gcodes, qs, r, k, None, scales)
    for j, qv in enumerate(qs):
        approx = opq.adc_desc(opq.preprocess_query(qv), codes, desc, scales)
        ws, wi = orc.topk_from_scores(approx, k)
        m = min(k, n)
        assert np.array_equal(bi[j, :m], wi[:m]) and np.array_equal(bs[j, :m], ws[:m]), j
        assert np.all(bi[j, m:] == 0xFFFFFFFF) and np.all(bs[j, m:] == np.iinfo(np.int64).min)
        s1, i1 = gpq.scan_topk(gcodes, qv, r, k, None, scales)
        assert np.array_equal(s1, bs[j]) and np.array_equal(i1, bi[j])
    # no descriptors at all
    g2 = mse.Codes(codes, None)
    s2, i2 = gpq.scan_topk_batch(g2, qs[:2], r, k)
    for j in range(2):
        ws, wi = orc.topk_from_scores(opq.asymmetric_dot_product(opq.preprocess_query(qs[j]), codes), k)
        m = min(k, n)
        assert np.array_equal(i2[j, :m], wi[:m]) and np.array_equal(s2[j, :m], ws[:m])


@pytest.mark.parametrize("n", [64 * 300, 64 * 97 + 13])
def test_group_maxima_equal_the_gathered_scores(gpu, mse, orc, n):
    """The select after the flat scan uses the r-th best GROUP MAXIMUM as a floor for the re-scored vectors, so the scan kernels
    (one query per pass, two queries per pass) and the gather kernel must produce the SAME i64 for a vector -- descriptor bias and
    contraction behaviour included.  Random tables (not from a codec: wide magnitudes, both signs), random codes and descriptors:
    every group maximum equals the maximum of the gathered scores of its 64 vectors, and both equal the oracle's."""
    rng = np.random.default_rng(n)
    cents, T, _, _ = make_pq(orc)
    opq, gpq = orc.PQ(cents, T, 18, D), mse.ProductQuantizer(cents, T, 18, D)
    codes = rng.integers(0, 256, size=(n, 64), dtype=np.uint8)
    desc = rng.integers(0, 256, size=(n, 4), dtype=np.uint8)
    gcodes = mse.Codes(codes, desc)
    luts = [(rng.standard_normal(64 * 256) * 10.0 ** rng.uniform(-4, 1, size=64 * 256)).astype(np.float32) for _ in range(2)]
    ids = np.arange(n, dtype=np.uint32)
    ng = (n + 63) // 64
    for scales in (None, np.array([0.5, -0.25, 3.0, 1e-3], np.float32) / np.float32(512)):
        want = []
        for lut in luts:
            g = gpq.adc_gather(gcodes, lut, ids, scales)                        # pq_adc_kernel, one vector per lane
            o = opq.adc_desc(lut, codes, desc, scales) if scales is not None else opq.asymmetric_dot_product(lut, codes)
            assert np.array_equal(g, o)
            pad = np.full(ng * 64, np.iinfo(np.int64).min, np.int64)
            pad[:n] = g
            want.append(pad.reshape(ng, 64).max(axis=1))
        one = [gpq.debug_group_max(gcodes, lut, None, scales) for lut in luts]
        two = gpq.debug_group_max(gcodes, luts[0], luts[1], scales)
        for j in range(2):
            assert np.array_equal(one[j], want[j]) and np.array_equal(two[j], want[j]), (j, scales is None)


def test_pq_four_queries_per_pass_is_certified_and_exact(gpu, mse, orc):
    """Batches of >= 4 queries go through the codes four per pass: a 12-bit integer nomination scan, the nominated groups re-scored
    in the reference's arithmetic, and a certificate that no excluded vector can reach the exact r-th score (pq.hip).  On codes with
    spread-out scores every query must be certified (the fast path really runs) and equal the oracle's ADC ranking; on codes with
    massive exact ties the certificate cannot hold (strict >), those queries are repeated through the exact scan, and the answer is
    still the oracle's."""
    rng = np.random.default_rng(91)
    cents, T, _, _ = make_pq(orc)
    opq, gpq = orc.PQ(cents, T, 18, D), mse.ProductQuantizer(cents, T, 18, D)
    n, r, k = 300_000, 120, 10
    codes = rng.integers(0, 256, size=(n, 64), dtype=np.uint8)
    desc = rng.integers(0, 256, size=(n, 4), dtype=np.uint8)
    scales = np.array([0.5, 0, -0.25, 0.125], np.float32) / np.float32(512)
    qs = (rng.standard_normal((9, D)) / np.sqrt(D)).astype(np.float32)
    for sc, ds in ((scales, desc), (None, None)):
        gcodes = mse.Codes(codes, ds)
        bs, bi = gpq.scan_topk_batch(gcodes, qs, r, k, None, sc)
        assert gpq.last_uncertified == 0
        for j, qv in enumerate(qs):
            lut = opq.preprocess_query(qv)
            approx = opq.adc_desc(lut, codes, ds, sc) if sc is not None else opq.asymmetric_dot_product(lut, codes)
            ws, wi = orc.topk_from_scores(approx, k)
            assert np.array_equal(bi[j], wi) and np.array_equal(bs[j], ws), j
    # a query whose table overflows f32 (entries +-inf): no integer table can bracket it -- that query is not certified and takes the
    # exact scan, the other three of its group are unaffected; r too large for the four-query scan's selection: the exact pair path
    wild = qs[:4].copy()
    wild[2] *= np.float32(3e38)
    gcodes = mse.Codes(codes, None)
    bs, bi = gpq.scan_topk_batch(gcodes, wild, r, k)
    assert gpq.last_uncertified == 1
    for j in range(4):
        s1, i1 = gpq.scan_topk(gcodes, wild[j], r, k)
        assert np.array_equal(bs[j], s1) and np.array_equal(bi[j], i1), j
    big_r = 1900
    bs, bi = gpq.scan_topk_batch(gcodes, qs[:4], big_r, k)
    assert gpq.last_uncertified == 0
    for j in range(4):
        ws, wi = orc.topk_from_scores(opq.asymmetric_dot_product(opq.preprocess_query(qs[j]), codes), k)
        assert np.array_equal(bi[j], wi) and np.array_equal(bs[j], ws), j
    # seven distinct code rows: thousands of vectors share the r-th score exactly
    tied = rng.integers(0, 256, size=(7, 64), dtype=np.uint8)[rng.integers(0, 7, size=n)]
    gt = mse.Codes(tied, None)
    bs, bi = gpq.scan_topk_batch(gt, qs[:4], r, k)
    assert gpq.last_uncertified == 4
    for j in range(4):
        ws, wi = orc.topk_from_scores(opq.asymmetric_dot_product(opq.preprocess_query(qs[j]), tied), k)
        assert np.array_equal(bi[j], wi) and np.array_equal(bs[j], ws), j
    # EIGHT per pass (8-bit tables) on the spread-out codes: certified and exact, like the first batch above (9 = 8 + 1 queries);
    # on the tied codes all eight are repeated through the exact scan -- answers still the oracle's -- and the handle then stays
    # with four per pass (12-bit tables), which a following batch of eight shows by being repeated as 4 + 4
    q8 = (rng.standard_normal((8, D)) / np.sqrt(D)).astype(np.float32)
    gcodes = mse.Codes(codes, desc)
    bs, bi = gpq.scan_topk_batch(gcodes, q8, r, k, None, scales)
    assert gpq.last_uncertified == 0
    for j in range(8):
        ws, wi = orc.topk_from_scores(opq.adc_desc(opq.preprocess_query(q8[j]), codes, desc, scales), k)
        assert np.array_equal(bi[j], wi) and np.array_equal(bs[j], ws), j
    bs, bi = gpq.scan_topk_batch(gt, q8, r, k)
    assert gpq.last_uncertified == 8
    for j in range(8):
        ws, wi = orc.topk_from_scores(opq.asymmetric_dot_product(opq.preprocess_query(q8[j]), tied), k)
        assert np.array_equal(bi[j], wi) and np.array_equal(bs[j], ws), j
    bs2, bi2 = gpq.scan_topk_batch(gcodes, q8, r, k, None, scales)       # four per pass now: same answers
    assert gpq.last_uncertified == 0
    bs, bi = gpq.scan_topk_batch(gcodes, q8[:4], r, k, None, scales)
    assert np.array_equal(bs2[:4], bs) and np.array_equal(bi2[:4], bi)


def test_pq_scan_full_size_1e8(gpu, mse, orc):
    """BASELINE.md's configs[4] size: 1e8 x 64-byte codes (+ 4 descriptor bytes), 6.8 GB in HBM.  The oracle cannot scan that in a
    test, so size-independent properties: planted best-possible vectors (the per-chunk argmax codes of a query) come back first,
    ties by lower id; every returned score is re-derived by the oracle from the returned id's code row; results are sorted; the
    k-th score bounds 2e5 sampled outsiders; pairs of queries sharing one pass equal the one-query-per-call answers."""
    n, r, k = 100_000_000, 200, 10
    rng = np.random.default_rng(77)
    cents, T, _, _ = make_pq(orc)
    opq, gpq = orc.PQ(cents, T, 18, D), mse.ProductQuantizer(cents, T, 18, D)
    block = rng.integers(0, 256, size=(1_000_000, 64), dtype=np.uint8)
    masks = rng.integers(0, 256, size=(100, 64), dtype=np.uint8)
    codes = np.empty((n, 64), np.uint8)
    for c in range(100):                                   # 100 distinct pseudo-random copies of one block: seconds, not minutes
        np.bitwise_xor(block, masks[c], out=codes[c * 1_000_000:(c + 1) * 1_000_000])
    desc = np.tile(rng.integers(0, 256, size=(1_000_000, 4), dtype=np.uint8), (100, 1))
    scales = np.array([0.5, 0, -0.25, 0], np.float32) / np.float32(512)
    qs = (rng.standard_normal((5, D)) / np.sqrt(D)).astype(np.float32)
    luts = [opq.preprocess_query(q) for q in qs]
    planted = [99_999_999, 64, 63, 50_000_001]            # group seams, the last vector, the middle
    best = np.argmax(luts[0].reshape(64, 256), axis=1).astype(np.uint8)
    for p in planted:
        codes[p] = best
        desc[p] = (255, 0, 0, 0)                          # the largest bias the scales allow
    gcodes = mse.Codes(codes, desc)
    bs, bi = gpq.scan_topk_batch(gcodes, qs, r, k, None, scales)
    assert gpq.last_uncertified <= 1          # query 0 has four planted vectors tied at the top, far above rank r; random codes certify
    assert sorted(bi[0, :4].tolist()) == sorted(planted) and bi[0, :4].tolist() == sorted(planted)   # equal scores: lower id first
    assert len(set(bs[0, :4].tolist())) == 1
    for j in range(5):
        rows = bi[j].astype(np.int64)
        assert np.all(rows < n) and len(set(rows.tolist())) == k
        assert np.array_equal(bs[j], opq.adc_desc(luts[j], codes[rows], desc[rows], scales))     # the oracle's score of that row
        assert np.all(bs[j, :-1] >= bs[j, 1:])
        s1, i1 = gpq.scan_topk(gcodes, qs[j], r, k, None, scales)                                 # one query per pass
        assert np.array_equal(s1, bs[j]) and np.array_equal(i1, bi[j])
    sample = rng.integers(0, n, 200_000)
    for j in (1, 4):
        sc = opq.adc_desc(luts[j], codes[sample], desc[sample], scales)
        assert np.all(sc[~np.isin(sample, bi[j])] <= bs[j, -1])


@pytest.mark.parametrize("n,d", [(0, 128), (1, 128), (777, 128), (3000, 1152), (20000, 256)])
def test_flat_index_matches_oracle(gpu, mse, orc, n, d):
    rng = np.random.default_rng(4)
    x = (rng.standard_normal((n, d)) / np.sqrt(d)).astype(np.float32)
    idx = mse.ScalarQuantizerIndex(d)
    for lo in range(0, n, 1024):                           # INDEX_ADD_BATCH (src/main.rs:815)
        idx.add(x[lo:lo + 1024])
    assert idx.ntotal() == n
    q = rng.standard_normal((11, d)).astype(np.float32)    # queries are NOT unit norm (common.rs:215-274)
    res = idx.search(q, 7)
    codes = orc.f16_bits(x)
    wd, wl = orc.index_search(codes, q, 7, order=0)
    assert np.array_equal(res.labels, wl)
    assert np.array_equal(res.distances, wd)
    if n >= 7:
        # FAISS's own (scalar, non-AVX2 build) summation order ranks this data identically
        _, l1 = orc.index_search(codes, q, 7, order=1)
        assert np.array_equal(res.labels, l1)
    assert np.all(res.labels[:, min(n, 7):] == -1)


def test_flat_index_at_config0_size(gpu, mse, orc):
    """BASELINE configs[0]: brute-force top-10 over 1e5 x 1152 through the in-memory index surface (src/main.rs:815-934):
    1024-row add batches, 11 un-normalised f32 queries, labels AND distances against the oracle's FAISS restatement."""
    rng = np.random.default_rng(40)
    n, d, k = 100_000, 1152, 10
    x = (rng.standard_normal((n, d)) / np.sqrt(d)).astype(np.float32)
    idx = mse.ScalarQuantizerIndex(d)
    for lo in range(0, n, 1024):
        idx.add(x[lo:lo + 1024])
    assert idx.ntotal() == n
    q = rng.standard_normal((11, d)).astype(np.float32)
    res = idx.search(q, k)
    codes = orc.f16_bits(x)
    wd, wl = orc.index_search(codes, q, k, order=0)
    assert np.array_equal(res.labels, wl) and np.array_equal(res.distances, wd)
    _, l1 = orc.index_search(codes, q, k, order=1)          # FAISS's scalar summation order ranks it identically
    assert np.array_equal(res.labels, l1)


def test_query_composed_by_get_total_embedding_end_to_end(gpu, mse, orc):
    """A16 (src/common.rs:215-274) in its place on the path: the clip server's fp16 rows (HIP text + image engines) are combined by
    get_total_embedding -- weights of both signs, a raw-embedding term, NO renormalisation -- and the sum is what the in-memory
    index is searched with (src/main.rs:941-953,900).  Composition against the oracle's restatement (bit-exact: the same f32
    multiply-adds in term order), then labels and distances of the search against the oracle's on the same query."""
    import io
    from PIL import Image
    from mse import siglip
    d = 1152
    eng = siglip.SiglipImageEngine.from_state_dict(siglip.synthetic_state_dict(dict(siglip.SO400M_384, depth=1)),
                                                   dict(siglip.SO400M_384, depth=1), max_batch=4)
    tcfg = dict(siglip.SO400M_TEXT, layers=1)
    teng = siglip.SiglipTextEngine.from_state_dict(siglip.synthetic_text_state_dict(tcfg), tcfg, max_batch=4)
    rng = np.random.default_rng(41)

    def bmp(seed):
        im = Image.fromarray(np.random.default_rng(seed).integers(0, 256, size=(384, 384, 3), dtype=np.uint8), "RGB")
        buf = io.BytesIO()
        im.save(buf, format="BMP")
        return buf.getvalue()

    tokens = {"cat": [5, 9, 2], "dog": [7, 3, 11, 2]}
    served = []

    def query_server(batch):                                  # the clip_server contract: a list of 2304-byte fp16 rows
        if "images" in batch:
            rows = eng.encode_bmp(batch["images"], out="f16")
        else:
            rows = teng.encode_text(siglip.pad_tokens([tokens[t] for t in batch["text"]]), out="f16")
        rows = [np.ascontiguousarray(r).view(np.uint16).tobytes() for r in rows]
        served.extend(rows)
        return rows

    raw = (rng.standard_normal(d) / np.sqrt(d)).astype(np.float32)
    terms = [{"image": bmp(1), "weight": 1.5}, {"text": "cat"}, {"text": "dog", "weight": -0.75}, {"image": bmp(2), "weight": -0.25},
             {"embedding": raw.tolist(), "weight": 0.5}]
    total = mse.get_total_embedding(terms, d, query_server)
    assert len(served) == 4 and all(len(r) == 2 * d for r in served)
    embs = np.stack([np.frombuffer(r, "<u2") for r in served])          # images first, then text (common.rs:252-266)
    weights = np.array([1.5, -0.25, 1.0, -0.75], np.float32)
    want = raw * np.float32(0.5) + orc.total_embedding(embs, weights)    # the oracle sums the served rows from zero: same terms,
    assert np.allclose(total, want, rtol=1e-6, atol=1e-7)                # different association of the raw-embedding term
    step = raw * np.float32(0.5)                                          # the reference's own order (:238-266), one f32 op at a time
    for e, w in zip(embs, weights):
        step = step + e.view(np.float16).astype(np.float32) * w
    assert np.array_equal(total, step)
    assert abs(float(np.linalg.norm(total)) - 1.0) > 1e-2                # a weighted sum, not re-normalised
    # the index: random rows plus the two image embeddings themselves
    n = 6000
    x = (rng.standard_normal((n, d)) / np.sqrt(d)).astype(np.float32)
    x[123] = mse.decode_fp16_buffer(served[0])
    x[4567] = mse.decode_fp16_buffer(served[1])
    idx = mse.ScalarQuantizerIndex(d)
    for lo in range(0, n, 1024):
        idx.add(x[lo:lo + 1024])
    res = idx.search(total[None, :], 10)
    wd, wl = orc.index_search(orc.f16_bits(x), total[None, :], 10, order=0)
    assert np.array_equal(res.labels, wl) and np.array_equal(res.distances, wd)
    assert res.labels[0, 0] == 123                                       # weight +1.5 on image 1 puts its own row first


def test_index_large_k_default_of_server(gpu, mse, orc):
    # handle_request uses k = 1000 by default (src/main.rs:952)
    rng = np.random.default_rng(5)
    d, n = 128, 5000
    x = (rng.standard_normal((n, d)) / np.sqrt(d)).astype(np.float32)
    idx = mse.ScalarQuantizerIndex(d)
    idx.add(x)
    q = rng.standard_normal((1, d)).astype(np.float32)
    res = idx.search(q, 1000)
    wd, wl = orc.index_search(orc.f16_bits(x), q, 1000, order=0)
    assert np.array_equal(res.labels, wl) and np.array_equal(res.distances, wd)


def test_greedy_search_matches_oracle(gpu, mse, orc):
    rng = np.random.default_rng(6)
    n, deg, L = 3000, 16, 64
    vecs = orc.gen_rows_f16(SEED_BASE, 0, n)
    # navigable-ish random graph: nearest few by a cheap projection + random long edges
    adj = rng.integers(0, n, size=(n, deg), dtype=np.uint32)
    degs = rng.integers(deg // 2, deg + 1, size=n).astype(np.uint32)
    searcher = mse.Searcher(mse.VectorList.from_f16s(vecs, D))
    graph = mse.IndexGraph(adj, degs)
    for qi in range(4):
        q = orc.gen_rows_f16(SEED_QUERY, qi, 1)[0]
        buf, dist = mse.greedy_search(searcher, 0, False, q, graph, L)
        obuf, odist = orc.greedy_search(vecs, adj, degs, 0, q, L)
        assert dist == odist
        assert np.array_equal(buf.ids, obuf.ids) and np.array_equal(buf.scores, obuf.scores)
    # base_vectors_only: ids >= breakpoint are never scored (OOD-DiskANN rule, lib.rs:196-199)
    q = orc.gen_rows_f16(SEED_QUERY, 9, 1)[0]
    buf, dist = mse.greedy_search(searcher, 0, True, q, graph, L, query_breakpoint=2000)
    obuf, odist = orc.greedy_search(vecs, adj, degs, 0, q, L, True, 2000)
    assert dist == odist and np.array_equal(buf.ids, obuf.ids)
    assert np.all(buf.ids[buf.ids != 0] < 2000)


def knn_graph(x, deg, rng, long_edges=4):
    """Small navigable graph for the beam-search tests: nearest neighbours by dot product + a few random edges."""
    n = len(x)
    s = x @ x.T
    np.fill_diagonal(s, -np.inf)
    near = np.argsort(-s, axis=1)[:, :deg - long_edges]
    adj = np.concatenate([near, rng.integers(0, n, size=(n, long_edges))], axis=1).astype(np.uint32)
    degs = rng.integers(deg - 2, deg + 1, size=n).astype(np.uint32)
    return adj, degs


@pytest.mark.parametrize("beamwidth,disable_pq,use_scales", [(1, False, True), (4, False, True), (3, True, True), (4, False, False)])
def test_disk_greedy_search_matches_oracle(gpu, mse, orc, beamwidth, disable_pq, use_scales):
    """query_disk_index::greedy_search (src/query_disk_index.rs:144-212): neighbour buffer, visited list in fetch order
    and both counters are bit-identical to the oracle, including the pre-buffer quirk for beamwidth > 1."""
    rng = np.random.default_rng(11)
    n, deg, L = 2500, 12, 48
    x = clustered_rows(orc, n, n_centres=32)
    base = orc.f16_bits(x)
    cents, T = train_pq(orc, x[:2000], iters=2)
    opq, gpq = orc.PQ(cents, T, 18, D), mse.ProductQuantizer(cents, T, 18, D)
    codes = opq.quantize_batch(orc.f16_to_f32(base))
    desc = rng.integers(0, 256, size=(n, 4), dtype=np.uint8)
    scales = (np.array([0.5, 0, -0.25, 1.0], np.float32) / np.float32(512)) if use_scales else None
    has_url = (rng.random(n) > 0.1).astype(np.uint8)
    adj, degs = knn_graph(x, deg, rng)
    searcher = mse.Searcher(mse.VectorList.from_f16s(base, D))
    gcodes = mse.Codes(codes, desc)
    graph = mse.IndexGraph(adj, degs)
    total_recall = 0
    for qi in range(3):
        qv = clustered_rows(orc, 1, n_centres=32, seed=100 + qi)[0]
        qh = orc.f16_bits(qv)
        lut = opq.preprocess_query(qv)
        start = int(rng.integers(0, n))
        res = mse.disk_greedy_search(searcher, gpq, gcodes, graph, start, qh, mse.QueryLUT(lut), scales, disable_pq, beamwidth,
                                     search_list=L, has_url=has_url)
        obuf, ovids, ovsc, ocm, opc = orc.disk_greedy_search(base, adj, degs, codes, desc, start, qh, lut, scales, disable_pq,
                                                              beamwidth, L, has_url)
        assert (res.cmps, res.pq_cmps) == (ocm, opc)
        assert np.array_equal(res.neighbour_buffer.ids, obuf.ids) and np.array_equal(res.neighbour_buffer.scores, obuf.scores)
        assert np.array_equal(res.visited_ids, ovids) and np.array_equal(res.visited_scores, ovsc)
        assert res.cmps >= len(res.visited_ids) > 0 and np.all(has_url[res.visited_ids] == 1)
        if disable_pq:
            assert res.pq_cmps == 0
        # the search is useful, not only self-consistent: visited list reaches most of the true top-10
        truth = orc.score_all(base, qh)
        if scales is not None:
            truth = truth + np.array([orc.descriptor_product(scales, desc, i) for i in range(n)])
        truth[has_url == 0] = np.iinfo(np.int64).min
        top = np.argsort(truth, kind="stable")[::-1][:10]
        total_recall += len(set(top.tolist()) & set(res.visited_ids.tolist()))
    assert total_recall >= 15, total_recall
    # errors: edge outside the index, start outside the index
    bad = adj.copy()
    bad[start, 0] = n + 5
    with pytest.raises(mse.MseError):
        mse.disk_greedy_search(searcher, gpq, gcodes, mse.IndexGraph(bad, degs), start, qh, mse.QueryLUT(lut))
    with pytest.raises(mse.MseError):
        mse.disk_greedy_search(searcher, gpq, gcodes, graph, n, qh, mse.QueryLUT(lut))


def test_select_shard_and_medioid(gpu, mse, orc):
    rng = np.random.default_rng(12)
    cents = rng.standard_normal((42, D)).astype(np.float32)          # kmeans.py:10 -> 42 shards
    for i in range(6):
        q = rng.standard_normal(D).astype(np.float32)
        assert mse.select_shard(cents, q) == orc.select_shard(cents, q)
    cents[17] = cents[5]                                             # exact tie: position_max_by_key keeps the LAST maximum
    q = (cents[5] * 3).astype(np.float32)
    assert mse.select_shard(cents, q) == orc.select_shard(cents, q) == 17
    for n in (1, 2, 777, 5000):
        vecs = orc.gen_rows_f16(SEED_BASE, 3, n)
        vl = mse.VectorList.from_f16s(vecs, D)
        assert mse.medioid(vl) == orc.medioid(vecs)
    dup = np.concatenate([vecs[:10], vecs[:10]])                     # every row twice: the later copy wins ties
    assert mse.medioid(mse.VectorList.from_f16s(dup, D)) == orc.medioid(dup) >= 10


def test_dedup_visited_matches_oracle(gpu, mse, orc):
    """query_disk_index.rs:482-527: near-duplicates (dot > 0.95 with an already kept row) are dropped, first one wins."""
    rng = np.random.default_rng(13)
    n = 900
    x = clustered_rows(orc, n, n_centres=40, noise=0.05)            # tight clusters: many pairs above 0.95
    x[100] = x[3]; x[500] = x[3]; x[501] = x[499]                     # exact duplicates as well
    base = orc.f16_bits(x)
    searcher = mse.Searcher(mse.VectorList.from_f16s(base, D))
    for m in (1, 65, 300):
        ids = rng.permutation(n)[:m].astype(np.uint32)
        keep = mse.dedup_visited(searcher, ids)
        want = orc.dedup_keep(base[ids]).astype(bool)
        assert np.array_equal(keep, want)
        assert keep[0] and (m < 65 or not keep.all())
    sims = x[ids][keep] @ x[ids][keep].T
    np.fill_diagonal(sims, 0)
    assert sims.max() <= 0.95 + 1e-3                                 # survivors are mutually dissimilar
    assert mse.dedup_visited(searcher, np.empty(0, np.uint32)).size == 0
    with pytest.raises(mse.MseError):
        mse.dedup_visited(searcher, np.array([n], np.uint32))


def test_index_packing_pieces(gpu, mse, orc):
    """score_model.rs:13-32 and dump_processor.rs:483-491 on the device against the oracle."""
    rng = np.random.default_rng(14)
    d, hdim, oc = D, 128, 3
    up = (rng.standard_normal((hdim, d)) / np.sqrt(d)).astype(np.float32)
    b = (0.1 * rng.standard_normal(hdim)).astype(np.float32)
    down = (rng.standard_normal((oc, hdim)) / np.sqrt(hdim)).astype(np.float32)
    x = clustered_rows(orc, 300)
    sm = mse.ScoreModel(up, b, down)
    got = sm.score_batch(x)
    want = orc.score_batch(up, b, down, x)
    assert got.shape == (300, oc) and np.allclose(got, want, rtol=1e-5, atol=1e-5)
    cdfs = np.sort(rng.standard_normal((4, 255)).astype(np.float32), axis=1)
    cdfs[2, 10:14] = cdfs[2, 10]                                         # a run of equal quantiles
    scores = np.concatenate([rng.standard_normal((1000, 4)).astype(np.float32), cdfs[:, :40].T,
                             np.full((1, 4), 1e9, np.float32), np.full((1, 4), -1e9, np.float32)])
    assert np.array_equal(mse.descriptor_buckets(cdfs, scores), orc.descriptor_buckets(cdfs, scores))
    with pytest.raises(mse.MseError):
        mse.descriptor_buckets(np.zeros((4, 300), np.float32), scores)


@pytest.mark.parametrize("beamwidth,disable_pq,use_scales", [(1, False, True), (4, False, True), (3, True, True), (8, False, False)])
def test_device_resident_beam_search_matches_oracle(gpu, mse, orc, beamwidth, disable_pq, use_scales):
    """The batched, GPU-resident search (one workgroup per query) returns, query by query, exactly what the oracle's
    restatement of query_disk_index::greedy_search returns: buffer, visited list in fetch order, both counters."""
    rng = np.random.default_rng(15)
    n, deg, L = 3000, 14, 64
    x = clustered_rows(orc, n, n_centres=32)
    base = orc.f16_bits(x)
    cents, T = train_pq(orc, x[:2000], iters=2)
    opq, gpq = orc.PQ(cents, T, 18, D), mse.ProductQuantizer(cents, T, 18, D)
    codes = opq.quantize_batch(orc.f16_to_f32(base))
    desc = rng.integers(0, 256, size=(n, 4), dtype=np.uint8)
    scales = (np.array([0.5, 0, -0.25, 1.0], np.float32) / np.float32(512)) if use_scales else None
    has_url = (rng.random(n) > 0.1).astype(np.uint8)
    adj, degs = knn_graph(x, deg, rng)
    adj[5, 3] = adj[5, 1]                                              # an id listed twice in one adjacency list
    searcher = mse.Searcher(mse.VectorList.from_f16s(base, D))
    gcodes = mse.Codes(codes, desc)
    dgraph = mse.DeviceGraph(mse.IndexGraph(adj, degs), has_url)
    nq = 9
    qs = clustered_rows(orc, nq, n_centres=32, seed=200)
    qh = orc.f16_bits(qs)
    luts = np.stack([opq.preprocess_query(q) for q in qs])
    starts = rng.integers(0, n, size=nq).astype(np.uint32)
    starts[0] = 5
    got = mse.disk_search_batch(searcher, gpq, gcodes, dgraph, starts, qh, luts, scales, disable_pq, beamwidth, search_list=L,
                                visited_cap=n)
    for i in range(nq):
        obuf, ovids, ovsc, ocm, opc = orc.disk_greedy_search(base, adj, degs, codes, desc, int(starts[i]), qh[i], luts[i], scales,
                                                              disable_pq, beamwidth, L, has_url)
        bi, bs, vi, vs, cm, pc = got[i]
        assert (cm, pc) == (ocm, opc), i
        assert np.array_equal(bi, obuf.ids) and np.array_equal(bs, obuf.scores), i
        assert np.array_equal(vi, ovids) and np.array_equal(vs, ovsc), i
    with pytest.raises(mse.MseError):
        mse.disk_search_batch(searcher, gpq, gcodes, dgraph, starts, qh, luts, scales, disable_pq, 9, search_list=L)
    with pytest.raises(mse.MseError):
        mse.disk_search_batch(searcher, gpq, gcodes, dgraph, np.full(nq, n, np.uint32), qh, luts, scales, disable_pq, 2, search_list=L)


@pytest.mark.parametrize("beamwidth,disable_pq,L,nq", [(4, True, 16, 40), (4, False, 16, 40), (2, False, 40, 24), (3, True, 40, 1100), (4, True, 12, 1100),
                                                       (4, True, 300, 1100)])
def test_beam_search_among_many_equal_scores_matches_oracle(gpu, mse, orc, beamwidth, disable_pq, L, nq):
    """Round 6 rewired what an iteration does when scores coincide: the flag is per iteration, only a LIVE newcomer's equalities (one a
    full list does not reject outright) send the iteration down the sequential insert path, rejected offers are dropped 64 at a time,
    only live newcomers are ranked, and an exactly scored search takes a fetched node's record from the list.  A base in which a third
    of the rows are exact copies of other rows (equal exact scores AND equal code bytes, hence equal ADC scores), short lists that
    fill up in the first iteration, duplicates inside adjacency lists: buffer, visited list in fetch order and both counters equal the
    oracle's for every query -- through the four-wave kernel (small batches, and every ADC-scored search) and the one-wave kernel
    (more than 1024 exactly scored queries per launch)."""
    rng = np.random.default_rng(77)
    n, deg = 2400, 16
    x = clustered_rows(orc, n, n_centres=12)
    src = rng.integers(0, n, size=n // 3)
    dst = rng.choice(n, size=n // 3, replace=False)
    x[dst] = x[src]                                                     # exact copies: ties everywhere
    base = orc.f16_bits(x)
    cents, T = train_pq(orc, x[:1500], iters=2)
    opq, gpq = orc.PQ(cents, T, 18, D), mse.ProductQuantizer(cents, T, 18, D)
    codes = opq.quantize_batch(orc.f16_to_f32(base))
    desc = rng.integers(0, 256, size=(n, 4), dtype=np.uint8)
    desc[dst] = desc[src]                                               # the copies tie with bias and without
    scales = np.array([0.5, 0, -0.25, 1.0], np.float32) / np.float32(512)
    adj, degs = knn_graph(x, deg, rng)
    adj[7, 5] = adj[7, 2]
    searcher = mse.Searcher(mse.VectorList.from_f16s(base, D))
    gcodes = mse.Codes(codes, desc)
    dgraph = mse.DeviceGraph(mse.IndexGraph(adj, degs))
    qs = clustered_rows(orc, nq, n_centres=12, seed=300)
    qs[::5] = x[rng.integers(0, n, size=len(qs[::5]))]                  # queries that ARE rows: their copies tie at the top of the list
    qh = orc.f16_bits(qs)
    check = range(nq) if nq <= 64 else range(0, nq, 9)
    luts = np.zeros((nq, 64 * 256), np.float32)
    if not disable_pq:
        luts = np.stack([opq.preprocess_query(q) for q in qs])
    starts = rng.integers(0, n, size=nq).astype(np.uint32)
    got = mse.disk_search_batch(searcher, gpq, gcodes, dgraph, starts, qh, luts, scales, disable_pq, beamwidth, search_list=L, visited_cap=n)
    replayed = 0
    for i in check:
        obuf, ovids, ovsc, ocm, opc = orc.disk_greedy_search(base, adj, degs, codes, desc, int(starts[i]), qh[i], luts[i], scales, disable_pq,
                                                              beamwidth, L, None)
        bi, bs, vi, vs, cm, pc = got[i]
        assert (cm, pc) == (ocm, opc), i
        assert np.array_equal(bi, obuf.ids) and np.array_equal(bs, obuf.scores), i
        assert np.array_equal(vi, ovids) and np.array_equal(vs, ovsc), i
        replayed += int(len(np.unique(bs)) < len(bs))
    assert replayed > 0                                                  # equal scores really sat inside the lists


@pytest.mark.parametrize("beamwidth,disable_pq,use_scales,L", [(4, True, False, 32), (2, False, True, 48), (4, True, True, 8)])
def test_request_path_in_one_call_matches_oracle(gpu, mse, orc, beamwidth, disable_pq, use_scales, L):
    """mse_disk_query_topk = entry node by the entry table + greedy_search + the visited records by exact score, first k
    (query_disk_index.rs:436-540): against the oracle's search from the oracle-chosen entry and a host sort of its visited list."""
    rng = np.random.default_rng(21)
    n, deg, k = 3000, 14, 10
    x = clustered_rows(orc, n, n_centres=32)
    base = orc.f16_bits(x)
    cents, T = train_pq(orc, x[:2000], iters=2)
    opq, gpq = orc.PQ(cents, T, 18, D), mse.ProductQuantizer(cents, T, 18, D)
    codes = opq.quantize_batch(orc.f16_to_f32(base))
    desc = rng.integers(0, 256, size=(n, 4), dtype=np.uint8)
    scales = (np.array([0.5, 0, -0.25, 1.0], np.float32) / np.float32(512)) if use_scales else None
    adj, degs = knn_graph(x, deg, rng)
    vecs = mse.VectorList.from_f16s(base, D)
    searcher = mse.Searcher(vecs)
    gcodes = mse.Codes(codes, desc)
    dgraph = mse.DeviceGraph(mse.IndexGraph(adj, degs))
    nq = 37
    qs = clustered_rows(orc, nq, n_centres=32, seed=300)
    qh = orc.f16_bits(qs)
    luts = np.stack([opq.preprocess_query(q) for q in qs])
    entry_ids = np.sort(rng.choice(n, 40, replace=False)).astype(np.uint32)
    with pytest.raises(mse.MseError):                                  # no entry table yet
        mse.disk_query_topk(searcher, gpq, gcodes, dgraph, qh, k, None, luts, scales, disable_pq, beamwidth, L)
    mse.set_entries(dgraph, vecs, entry_ids)
    ids, scores, stats = mse.disk_query_topk(searcher, gpq, gcodes, dgraph, qh, k, None, luts, scales, disable_pq, beamwidth, L)
    assert ids.shape == (nq, k) and scores.shape == (nq, k)
    # the oracle's entry: the entry row with the largest exact dot product (lower row on a tie), then its search and a sort
    _, best = orc.bruteforce_topk(base[entry_ids], qh, 1)
    starts = entry_ids[best[:, 0]]
    for i in range(nq):
        _, ovids, ovsc, ocm, opc = orc.disk_greedy_search(base, adj, degs, codes, desc, int(starts[i]), qh[i], luts[i], scales,
                                                          disable_pq, beamwidth, L, None)
        order = np.array(sorted(range(len(ovids)), key=lambda j: (-int(ovsc[j]), int(ovids[j]))), np.int64)   # score descending, id ascending on equal scores
        want_ids = np.full(k, 0xFFFFFFFF, np.uint32)
        want_sc = np.full(k, np.iinfo(np.int64).min, np.int64)
        m = min(k, len(order))
        want_ids[:m] = ovids[order[:m]]
        want_sc[:m] = ovsc[order[:m]]
        assert np.array_equal(ids[i], want_ids) and np.array_equal(scores[i], want_sc), i
        assert (int(stats["cmps"][i]), int(stats["pq_cmps"][i]), int(stats["n_visited"][i])) == (ocm, opc, len(ovids)), i
    # queries that are already on the device (a pointer instead of a host array): the same answer
    import torch
    qd = torch.from_numpy(qh.view(np.int16).copy()).cuda()
    ids_d, scores_d, _ = mse.disk_query_topk(searcher, gpq, gcodes, dgraph, (qd.data_ptr(), nq), k, None, luts, scales, disable_pq, beamwidth, L)
    assert np.array_equal(ids_d, ids) and np.array_equal(scores_d, scores)
    # given start nodes: the same call without the entry step; and the batched search's own visited lists agree
    ids2, scores2, _ = mse.disk_query_topk(searcher, gpq, gcodes, dgraph, qh, k, starts, luts, scales, disable_pq, beamwidth, L)
    assert np.array_equal(ids2, ids) and np.array_equal(scores2, scores)
    res = mse.disk_search_batch(searcher, gpq, gcodes, dgraph, starts, qh, luts, scales, disable_pq, beamwidth, search_list=L,
                                visited_cap=n, as_arrays=True)
    assert np.array_equal(mse.topk_of_visited(res, k), ids.astype(np.int64))
    # k beyond what a search visits: padded rows
    ids3, scores3, st3 = mse.disk_query_topk(searcher, gpq, gcodes, dgraph, qh[:3], 200, starts[:3], luts[:3],
                                             None if scales is None else scales, disable_pq, beamwidth, L)
    for i in range(3):
        nv = int(st3["n_visited"][i])
        assert np.all(ids3[i, nv:] == 0xFFFFFFFF) and np.all(scores3[i, nv:] == np.iinfo(np.int64).min) and np.all(ids3[i, :min(nv, 200)] != 0xFFFFFFFF)
    with pytest.raises(mse.MseError):
        mse.set_entries(dgraph, vecs, np.array([n], np.uint32))


def test_request_path_from_several_threads(gpu, mse, orc):
    """Four request threads, each with its own searcher, call mse_disk_query_topk at once (entry searchers come from the graph's
    pool): every thread gets what the call returns when made alone."""
    import threading
    rng = np.random.default_rng(22)
    n, deg, k, L = 4000, 16, 10, 24
    x = clustered_rows(orc, n, n_centres=40)
    base = orc.f16_bits(x)
    adj, degs = knn_graph(x, deg, rng)
    vecs = mse.VectorList.from_f16s(base, D)
    dgraph = mse.DeviceGraph(mse.IndexGraph(adj, degs))
    mse.set_entries(dgraph, vecs, np.sort(rng.choice(n, 64, replace=False)).astype(np.uint32))
    qsets = [orc.f16_bits(clustered_rows(orc, 300 + 17 * t, n_centres=40, seed=400 + t)) for t in range(4)]
    ref = mse.Searcher(vecs)
    want = [mse.disk_query_topk(ref, None, None, dgraph, q, k, None, None, None, True, 4, L) for q in qsets]
    got, errs = [None] * 4, []

    def worker(t):
        try:
            s = mse.Searcher(vecs)
            for _ in range(5):
                got[t] = mse.disk_query_topk(s, None, None, dgraph, qsets[t], k, None, None, None, True, 4, L)
            s.close()
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    th = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    for t in range(4):
        assert np.array_equal(got[t][0], want[t][0]) and np.array_equal(got[t][1], want[t][1])
        assert np.array_equal(got[t][2]["cmps"], want[t][2]["cmps"])


@pytest.mark.parametrize("disable_pq", [False, True])
def test_device_resident_beam_search_from_f32_queries(gpu, mse, orc, disable_pq):
    """f32 queries in: the f16 copies (RNE) and the distance tables are made on the device (query_disk_index.rs:475-477);
    the result equals the call that is handed f16 queries and tables made one by one."""
    rng = np.random.default_rng(16)
    n, deg, L, nq = 2500, 12, 48, 11
    x = clustered_rows(orc, n, n_centres=24)
    base = orc.f16_bits(x)
    cents, T = train_pq(orc, x[:1500], iters=2)
    opq, gpq = orc.PQ(cents, T, 18, D), mse.ProductQuantizer(cents, T, 18, D)
    codes = opq.quantize_batch(orc.f16_to_f32(base))
    adj, degs = knn_graph(x, deg, rng)
    searcher = mse.Searcher(mse.VectorList.from_f16s(base, D))
    gcodes = mse.Codes(codes, None)
    dgraph = mse.DeviceGraph(mse.IndexGraph(adj, degs))
    qs = (clustered_rows(orc, nq, n_centres=24, seed=300) * np.float32(1.7)).astype(np.float32)   # queries need not be unit norm
    starts = rng.integers(0, n, size=nq).astype(np.uint32)
    luts = np.stack([gpq.preprocess_query(q).table for q in qs])
    want = mse.disk_search_batch(searcher, gpq, gcodes, dgraph, starts, orc.f16_bits(qs), luts, None, disable_pq, 3, search_list=L, visited_cap=n)
    got = mse.disk_search_batch(searcher, gpq, gcodes, dgraph, starts, qs, None, None, disable_pq, 3, search_list=L, visited_cap=n)
    for w, g_ in zip(want, got):
        assert all(np.array_equal(a, b) for a, b in zip(w[:4], g_[:4])) and w[4:] == g_[4:]
    # and against the oracle for one query
    obuf, ovids, ovsc, ocm, opc = orc.disk_greedy_search(base, adj, degs, codes, None, int(starts[0]), orc.f16_bits(qs[0]),
                                                          opq.preprocess_query(qs[0]), None, disable_pq, 3, L, None)
    # one table builder on the device (pq_transform_kernel + pq_lut_kernel, single or batched launch) and it is bit-equal to the
    # oracle's (test_codec_matches_oracle), so ADC mode must agree with the oracle outright as well: buffer, visited list, counters
    assert np.array_equal(np.asarray(luts[0]).reshape(-1), opq.preprocess_query(qs[0]).reshape(-1))
    bi, bs, vi, vs, cm, pc = got[0]
    assert np.array_equal(bi, obuf.ids) and np.array_equal(bs, obuf.scores)
    assert np.array_equal(vi, ovids) and np.array_equal(vs, ovsc)
    assert (cm, pc) == (ocm, opc)


def test_long_batches_go_through_in_pieces(gpu, mse, orc, monkeypatch):
    """A batch whose visited sets would not fit the budget is cut into pieces; the result does not depend on the cut."""
    rng = np.random.default_rng(17)
    n, deg, L, nq = 2000, 10, 32, 37
    x = clustered_rows(orc, n, n_centres=16)
    base = orc.f16_bits(x)
    adj, degs = knn_graph(x, deg, rng)
    searcher = mse.Searcher(mse.VectorList.from_f16s(base, D))
    cents, T, _, _ = make_pq(orc)
    gpq = mse.ProductQuantizer(cents, T, 18, D)
    gcodes = mse.Codes(rng.integers(0, 256, size=(n, 64), dtype=np.uint8), None)
    dgraph = mse.DeviceGraph(mse.IndexGraph(adj, degs))
    qs = orc.f16_bits(clustered_rows(orc, nq, n_centres=16, seed=400))
    starts = rng.integers(0, n, size=nq).astype(np.uint32)
    luts = np.stack([gpq.preprocess_query(orc.f16_to_f32(q)).table for q in qs])
    whole = mse.disk_search_batch(searcher, gpq, gcodes, dgraph, starts, qs, luts, None, False, 2, search_list=L, visited_cap=256)
    bg = mse.BuildGraph(n, deg, mse.IndexGraph(adj, degs))
    whole_ram = bg.search_batch(searcher, starts, qs, L)
    monkeypatch.setenv("MSE_VISITED_BUDGET_KB", "3")     # 250 B (in-RAM) / 500 B (disk variant) per query -> pieces of 12 / 6
    cut = mse.disk_search_batch(searcher, gpq, gcodes, dgraph, starts, qs, luts, None, False, 2, search_list=L, visited_cap=256)
    cut_ram = bg.search_batch(searcher, starts, qs, L)
    for a, b in zip(whole, cut):
        assert all(np.array_equal(u, v) for u, v in zip(a[:4], b[:4])) and a[4:] == b[4:]
    for a, b in zip(whole_ram, cut_ram):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]


@pytest.mark.parametrize("beamwidth,disable_pq", [(4, False), (8, True)])
def test_device_resident_beam_search_wide_lists(gpu, mse, orc, beamwidth, disable_pq):
    """Merged indexes carry the union of a point's lists from its shards -- up to 2 R = 128 neighbours (dump_processor.rs:282-291).
    Lists of up to 100 ids, with an id repeated across the two 64-neighbour rounds, beam 8 x 100 newcomers per iteration."""
    rng = np.random.default_rng(23)
    n, deg, L, nq = 2500, 100, 80, 7
    x = clustered_rows(orc, n, n_centres=10)
    base = orc.f16_bits(x)
    cents, T = train_pq(orc, x[:1500], iters=1)
    opq, gpq = orc.PQ(cents, T, 18, D), mse.ProductQuantizer(cents, T, 18, D)
    codes = opq.quantize_batch(orc.f16_to_f32(base))
    adj, degs = knn_graph(x, deg, rng, long_edges=10)
    degs[:] = rng.integers(60, deg + 1, size=n)
    adj[9, 70] = adj[9, 3]                                             # repeated across the two rounds
    adj[9, 71] = adj[9, 70]                                            # and inside the second round
    searcher = mse.Searcher(mse.VectorList.from_f16s(base, D))
    gcodes = mse.Codes(codes, None)
    dgraph = mse.DeviceGraph(mse.IndexGraph(adj, degs))
    qs = clustered_rows(orc, nq, n_centres=10, seed=500)
    qh = orc.f16_bits(qs)
    luts = np.stack([opq.preprocess_query(q) for q in qs])
    starts = rng.integers(0, n, size=nq).astype(np.uint32)
    starts[0] = 9
    got = mse.disk_search_batch(searcher, gpq, gcodes, dgraph, starts, qh, luts, None, disable_pq, beamwidth, search_list=L, visited_cap=n)
    for i in range(nq):
        obuf, ovids, ovsc, ocm, opc = orc.disk_greedy_search(base, adj, degs, codes, None, int(starts[i]), qh[i], luts[i], None,
                                                              disable_pq, beamwidth, L, None)
        bi, bs, vi, vs, cm, pc = got[i]
        assert (cm, pc) == (ocm, opc), i
        assert np.array_equal(bi, obuf.ids) and np.array_equal(bs, obuf.scores), i
        assert np.array_equal(vi, ovids) and np.array_equal(vs, ovsc), i


@pytest.mark.parametrize("bits", [None, "12"])
def test_visited_sets_as_hash_tables(gpu, mse, orc, bits, monkeypatch):
    """Large indexes keep their visited sets in open-addressing tables instead of one bit per node (visited_set.h).  Forced here
    on a small index: same results; with a 4096-slot table the searches outgrow it and the call falls back to bit maps."""
    rng = np.random.default_rng(29)
    n, deg, L, nq = 6000, 40, 100, 12
    x = clustered_rows(orc, n, n_centres=12)
    base = orc.f16_bits(x)
    adj, degs = knn_graph(x, deg, rng, long_edges=6)
    searcher = mse.Searcher(mse.VectorList.from_f16s(base, D))
    dgraph = mse.DeviceGraph(mse.IndexGraph(adj, degs))
    bg = mse.BuildGraph(n, deg, mse.IndexGraph(adj, degs))
    qs = orc.f16_bits(clustered_rows(orc, nq, n_centres=12, seed=600))
    starts = rng.integers(0, n, size=nq).astype(np.uint32)
    want = mse.disk_search_batch(searcher, None, None, dgraph, starts, qs, None, None, True, 4, search_list=L, visited_cap=n)
    want_ram = bg.search_batch(searcher, starts, qs, L)
    monkeypatch.setenv("MSE_VISITED_MODE", "hash")
    if bits:
        monkeypatch.setenv("MSE_VISITED_TABLE_BITS", bits)
    got = mse.disk_search_batch(searcher, None, None, dgraph, starts, qs, None, None, True, 4, search_list=L, visited_cap=n)
    got_ram = bg.search_batch(searcher, starts, qs, L)
    for a, b in zip(want, got):
        assert all(np.array_equal(u, v) for u, v in zip(a[:4], b[:4])) and a[4:] == b[4:]
    for a, b in zip(want_ram, got_ram):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]
    assert max(r[4] for r in want) * 20 > 2048 or bits is None      # the small table really is outgrown (fetches x ~20 fresh ids)


def test_index_directory_is_the_front_door_of_the_beam_search(gpu, mse, orc, tmp_path):
    """an index directory written the way dump-processor packs it (index.bin records, code files, header),
    opened with DiskIndex, moved to HBM with to_device(), searched with the GPU-resident beam search: equal to the oracle's
    greedy_search over the arrays the directory was written from, including the records that lost their URL (graph-only nodes)."""
    from mse import disk_index as di
    rng = np.random.default_rng(23)
    n, deg, L, nq = 1200, 10, 40, 6
    x = clustered_rows(orc, n, n_centres=16)
    base = orc.f16_bits(x)
    cents, T = train_pq(orc, x[:800], iters=2)
    opq = orc.PQ(cents, T, 18, D)
    codes = opq.quantize_batch(orc.f16_to_f32(base))
    desc = rng.integers(0, 256, size=(n, 2), dtype=np.uint8)
    adj, degs = knn_graph(x, deg, rng)
    urls = ["https://example.org/%d" % i if i % 7 else "" for i in range(n)]            # every seventh record is a dead node
    hdr = di.IndexHeader([(x[:50].mean(axis=0).astype(np.float32), 3)], 0, 0, di.RECORD_PAD_SIZE,
                         {"centroids": cents.reshape(-1), "transform": T.reshape(-1), "n_dims_per_code": 18, "n_dims": D},
                         [np.linspace(0, 1, 5).astype(np.float32)] * 2)
    ents = ({"vector": base[i], "vertices": adj[i, :degs[i]], "id": i, "timestamp": 1_700_000_000 + i, "dimensions": (100 + i, 50),
             "scores": np.array([0.1, 0.2], np.float32), "url": urls[i], "shards": np.array([0], np.uint32)} for i in range(n))
    di.write_index(str(tmp_path), hdr, ents, codes, desc, encode_entry=di.UNPINNED_BITCODE06_ENCODE)
    idx = di.DiskIndex(str(tmp_path), decode_entry=di.UNPINNED_BITCODE06_DECODE)
    assert idx.header.count == n and idx.header.dead_count == 0       # empty URLs given by the caller are not "dead" by overflow
    vl, dgraph, gcodes, graph, got_urls = idx.to_device()
    assert got_urls == urls and np.array_equal(vl.rows(0, n), base)
    assert np.array_equal(graph.deg, degs) and all(np.array_equal(graph.adj[i, :degs[i]], adj[i, :degs[i]]) for i in range(n))
    gpq = idx.header.product_quantizer()
    searcher = mse.Searcher(vl)
    qs = clustered_rows(orc, nq, n_centres=16, seed=77)
    starts = np.full(nq, 3, np.uint32)                                                   # the shard's medioid from the header
    scales = np.array([0.5, -0.25], np.float32) / np.float32(512)
    got = mse.disk_search_batch(searcher, gpq, gcodes, dgraph, starts, qs, None, scales, False, 3, search_list=L, visited_cap=n)
    has_url = np.array([1 if u else 0 for u in urls], np.uint8)
    for i in range(nq):
        obuf, ovids, ovsc, ocm, opc = orc.disk_greedy_search(base, graph.adj, degs, codes, desc, 3, orc.f16_bits(qs[i]),
                                                              opq.preprocess_query(qs[i]), scales, False, 3, L, has_url)
        bi, bs, vi, vs, cm, pc = got[i]
        assert np.array_equal(bi, obuf.ids) and np.array_equal(bs, obuf.scores) and (cm, pc) == (ocm, opc)
        assert np.array_equal(vi, ovids) and np.array_equal(vs, ovsc)
        assert all(urls[int(v)] for v in vi)                                             # dead nodes are traversed, never returned


@pytest.mark.parametrize("n,n_valid,with_desc,per_pass", [(64 * 37 + 5, 4, True, 4), (1000, 3, False, 4), (16, 1, True, 4), (64 * 300, 4, True, 4),
                                                          (64 * 41 + 9, 8, True, 8), (3000, 5, False, 8), (64 * 300, 8, True, 8)])
def test_pq4_matrix_core_scan_equals_integer_sums(gpu, mse, orc, n, n_valid, with_desc, per_pass):
    """The nomination scan (pq_scan64x4_kernel: conflict-free rotated gathers, sums on the matrix cores; four queries per pass with
    12-bit tables or eight with 8-bit tables) against plain integer arithmetic: the tables are rebuilt on the host from the kernel's
    own formula (e = rint((lut - lo_c) / delta), descriptor chunks rint((sc v - min(0, 255 sc)) / delta)) and every group maximum
    must equal max over the group's vectors of sum_c e[c][code_c] (+ descriptor entries)."""
    import ctypes as C
    from mse import ffi
    rng = np.random.default_rng(n + n_valid)
    cents, T, dpc, d = make_pq(orc)
    pq = mse.ProductQuantizer(cents, T, dpc, d)
    codes = rng.integers(0, 256, (n, 64), dtype=np.uint8)
    codes[: min(n, 7)] = 255                                   # extreme codes in the first rows
    desc = rng.integers(0, 256, (n, 4), dtype=np.uint8)
    gc = mse.Codes(codes, desc)
    luts = (rng.standard_normal((per_pass, 64, 256)) * rng.uniform(0.01, 0.3, (per_pass, 64, 1))).astype(np.float32)
    luts[1, 3] = 0.25                                          # a chunk with zero range
    scales = np.array([0.5, 0, -0.25, 0.125], np.float32) / np.float32(512) if with_desc else None
    code_max, desc_max = (4095.0, 16383.0) if per_pass == 4 else (255.0, 16383.0)
    ng = (n + 63) // 64
    out = np.zeros((per_pass, ng), np.uint32)
    params = np.zeros((per_pass, 4), np.float64)
    ffi.check(ffi.lib().mse_debug_pq4_group_max(pq._h, gc._h, luts.ctypes.data_as(ffi.f32p),
                                                scales.ctypes.data_as(ffi.f32p) if with_desc else None, n_valid, per_pass,
                                                out.ctypes.data_as(ffi.u32p), params.ctypes.data_as(C.POINTER(C.c_double))))
    for j in range(per_pass):
        if j >= n_valid:
            assert params[j, 3] == 0 and np.all(out[j] == 0)
            continue
        lut = luts[j].astype(np.float64)
        lo, hi = lut.min(axis=1), lut.max(axis=1)
        delta = (hi - lo).max() / code_max
        c_sum = 0.0
        for c in range(64):
            c_sum += lo[c]
        if with_desc:
            for sc in scales.astype(np.float64):
                delta = max(delta, abs(sc) * 255.0 / desc_max)
                c_sum += min(0.0, sc * 255.0)
        delta = max(delta, 1e-300)
        assert params[j, 0] == delta and params[j, 1] == c_sum and params[j, 3] == 1
        inv = 1.0 / delta
        e = np.clip(np.rint((lut - lo[:, None]) * inv), 0, code_max).astype(np.int64)         # [64][256]
        sums = e[np.arange(64)[None, :], codes.astype(np.int64)].sum(axis=1)                  # [n]
        if with_desc:
            for dd, sc in enumerate(scales.astype(np.float64)):
                ed = np.clip(np.rint((sc * np.arange(256.0) - min(0.0, sc * 255.0)) * inv), 0, desc_max).astype(np.int64)
                sums = sums + ed[desc[:, dd].astype(np.int64)]
        pad = np.zeros(ng * 64, np.int64)
        pad[:n] = sums
        want = pad.reshape(ng, 64).max(axis=1)
        assert np.array_equal(out[j].astype(np.int64), want), (j, np.flatnonzero(out[j] != want)[:5])
        # the certificate's premise: the reference-order score of every vector lies within eps of delta * S + C
        if j == 0 and n <= 4000:
            adc = np.zeros(n, np.float32)
            for c in range(64):
                adc = (adc + luts[j][c][codes[:, c]]).astype(np.float32)
            x = adc.astype(np.float64)
            if with_desc:
                x = x + (scales.astype(np.float64)[None, :] * desc.astype(np.float64)).sum(axis=1)
            assert np.all(np.abs(x - (delta * sums + c_sum)) <= params[j, 2] * (1 + 1e-6) + 4e-9 * 6)


def test_codes_quantized_from_resident_rows_equal_quantize_batch(gpu, mse, orc):
    """mse_codes_quantize_base: the PQ codes of rows that already live in HBM (f16 -> f32 widening, transform, first-max argmax per
    chunk, all on the device) equal quantize_batch's on the same rows (vector.rs:331-364), and a scan over them answers the same."""
    n = 70000                                                   # more than one 65536-row step
    cents, T, dpc, d = make_pq(orc)
    rng = np.random.default_rng(12)
    base = orc.f16_bits((rng.standard_normal((n, d)) / np.sqrt(d)).astype(np.float32))
    pq = mse.ProductQuantizer(cents, T, dpc, d)
    vl = mse.VectorList.from_f16s(base, d)
    want = np.concatenate([pq.quantize_batch(orc.f16_to_f32(base[s:s + 8192])) for s in range(0, n, 8192)])
    assert np.array_equal(want[:2000], orc.PQ(cents, T, dpc, d).quantize_batch(orc.f16_to_f32(base[:2000])))
    desc = rng.integers(0, 256, (n, 4), dtype=np.uint8)
    dev_codes = mse.Codes.quantize_base(pq, vl, desc)
    host_codes = mse.Codes(want, desc)
    assert len(dev_codes) == n
    q = (rng.standard_normal((5, d)) / np.sqrt(d)).astype(np.float32)
    scales = np.array([0.5, 0, -0.25, 0.125], np.float32) / np.float32(512)
    s = mse.Searcher(vl)
    a = pq.scan_topk_batch(dev_codes, q, 100, 10, s, scales)
    b = pq.scan_topk_batch(host_codes, q, 100, 10, s, scales)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    # ADC-only answers expose the codes themselves: identical rankings and scores over all 70000 vectors' best 100
    a = pq.scan_topk_batch(dev_codes, q, 100, 100, None, scales)
    b = pq.scan_topk_batch(host_codes, q, 100, 100, None, scales)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_entry_by_shard_centroids_is_the_reference_rule(gpu, mse, orc):
    """mse_graph_set_entry_centroids: the start node is the medioid of the shard that orc.select_shard picks (f32 dot summed in f64,
    scale_dot_result_f64, LAST maximum on ties -- src/query_disk_index.rs:254-256,447-450), for f32 queries and for f16 queries
    (widened exactly), through the batch call; results equal the call that is given those start nodes."""
    rng = np.random.default_rng(41)
    n, deg, S, nq, k, L = 2500, 12, 42, 45, 10, 24
    x = clustered_rows(orc, n, n_centres=24)
    base = orc.f16_bits(x)
    adj, degs = knn_graph(x, deg, rng)
    vecs = mse.VectorList.from_f16s(base, D)
    searcher = mse.Searcher(vecs)
    dgraph = mse.DeviceGraph(mse.IndexGraph(adj, degs))
    centroids = (x[rng.choice(n, S, replace=False)] * np.float32(0.8) + rng.standard_normal((S, D)).astype(np.float32) * np.float32(0.01)).astype(np.float32)
    centroids[40] = centroids[3]                                       # equal keys for every query: shard 40 must win over shard 3
    medioids = rng.choice(n, S, replace=False).astype(np.uint32)
    mse.set_entry_centroids(dgraph, centroids, medioids)
    qs = (clustered_rows(orc, nq, n_centres=24, seed=310) * np.float32(1.1)).astype(np.float32)
    qs[0] = centroids[3]
    qh = orc.f16_bits(qs)
    for q_in, q_for_shard in ((qs, qs), (qh, orc.f16_to_f32(qh))):
        shards = np.array([orc.select_shard(centroids, q_for_shard[i]) for i in range(nq)])
        assert shards[0] == 40 and len(set(shards.tolist())) > 3
        got = mse.disk_query_topk(searcher, None, None, dgraph, q_in, k, None, None, None, True, 2, L)
        want = mse.disk_query_topk(searcher, None, None, dgraph, qh, k, medioids[shards], None, None, True, 2, L)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
        assert np.array_equal(got[2]["cmps"], want[2]["cmps"])
    with pytest.raises(mse.MseError):
        mse.set_entry_centroids(dgraph, centroids, np.full(S, n, np.uint32))
    # a row table replaces the centroid table (and the other way round)
    mse.set_entries(dgraph, vecs, medioids)
    _, best = orc.bruteforce_topk(base[medioids], qh, 1)
    got = mse.disk_query_topk(searcher, None, None, dgraph, qh, k, None, None, None, True, 2, L)
    want = mse.disk_query_topk(searcher, None, None, dgraph, qh, k, medioids[best[:, 0]], None, None, True, 2, L)
    assert np.array_equal(got[0], want[0])


def test_device_queries_written_by_another_stream(gpu, mse, orc):
    """Device-resident queries produced on ANOTHER stream (the text tower's): mse_searcher_wait_stream orders the searcher's copy after
    the producer without a host synchronisation.  The producer stream is kept busy for tens of milliseconds before it writes the
    queries, so an unordered copy would read the zeros that were there before."""
    import torch
    rng = np.random.default_rng(42)
    n, deg, nq, k, L = 3000, 12, 40, 10, 24
    x = clustered_rows(orc, n, n_centres=24)
    base = orc.f16_bits(x)
    adj, degs = knn_graph(x, deg, rng)
    vecs = mse.VectorList.from_f16s(base, D)
    searcher = mse.Searcher(vecs)
    dgraph = mse.DeviceGraph(mse.IndexGraph(adj, degs))
    mse.set_entries(dgraph, vecs, np.sort(rng.choice(n, 50, replace=False)).astype(np.uint32))
    qh = orc.f16_bits(clustered_rows(orc, nq, n_centres=24, seed=311))
    want = mse.disk_query_topk(searcher, None, None, dgraph, qh, k, None, None, None, True, 2, L)
    src = torch.from_numpy(qh.view(np.int16).copy()).cuda()
    qd = torch.zeros_like(src)
    a = torch.randn(4096, 4096, device="cuda")
    torch.cuda.synchronize()
    producer = torch.cuda.Stream()
    with torch.cuda.stream(producer):
        for _ in range(60):
            a = a @ a * 1e-3                                           # tens of milliseconds of work ahead of the write
        qd.copy_(src)
    searcher.wait_stream(producer.cuda_stream)
    got = mse.disk_query_topk(searcher, None, None, dgraph, (qd.data_ptr(), nq), k, None, None, None, True, 2, L)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    torch.cuda.synchronize()


def test_text_embedding_handed_to_the_search_on_the_device(gpu, mse, orc):
    """The request handler embeds the query text and searches with the embedding (src/query_disk_index.rs:345-381,436-540).
    mse_siglip_text_encode_dev leaves the tower's f16 rows on the device and waits for nothing; mse_searcher_wait_stream orders the
    search behind the engine's stream: the answers are those of the host round trip (features to the host, f16 host queries), for
    one request and for a batch, and the device rows are the rows mse_siglip_text_encode returns."""
    from mse import siglip
    from oracle import siglip_ref as ref
    rng = np.random.default_rng(77)
    n, deg, k, L = 3000, 12, 10, 24
    x = clustered_rows(orc, n, n_centres=24)
    base = orc.f16_bits(x)
    adj, degs = knn_graph(x, deg, rng)
    vecs = mse.VectorList.from_f16s(base, D)
    searcher = mse.Searcher(vecs)
    dgraph = mse.DeviceGraph(mse.IndexGraph(adj, degs))
    mse.set_entries(dgraph, vecs, np.sort(rng.choice(n, 50, replace=False)).astype(np.uint32))
    cfg = dict(ref.TEXT_CONFIG, layers=2)
    eng = siglip.SiglipTextEngine.from_state_dict(ref.synthetic_text_weights(cfg), dict(siglip.SO400M_TEXT, layers=2), max_batch=16)
    tok = ref.synthetic_tokens(16, cfg).numpy()
    for b in (1, 16, 5):
        rows = eng.encode_text(tok[:b], out="f16")
        want = mse.disk_query_topk(searcher, None, None, dgraph, rows, k, None, None, None, True, 2, L)
        _, p16, stream = eng.encode_text_device(tok[:b])
        searcher.wait_stream(stream)
        got = mse.disk_query_topk(searcher, None, None, dgraph, (p16, b), k, None, None, None, True, 2, L)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
        # the oracle's search for the same f16 rows, from the same start node rule (exact top-1 over the entry rows)
        assert got[0].shape == (b, k) and np.all(got[1][:, 0] >= got[1][:, -1])
    eng.close()


def test_search_measurement_hook_counts_what_the_searches_gathered(gpu, mse, orc):
    """mse_searcher_beam_timing (bench.py's gather roofline): the kernel's own totals agree with the per-query counters the call returns
    -- fetched nodes = sum of cmps; with ADC-scored neighbours every neighbour that entered a pre-buffer is one code gather and only the
    fetched nodes are scored exactly; with exactly scored neighbours every one of them is a row gather too -- and the hook changes no
    answer.  Iterations >= fetched nodes / beam; the sequential insert path is the exception, not the rule."""
    rng = np.random.default_rng(91)
    n, deg, nq, k, L, beam = 4000, 12, 48, 10, 40, 4
    x = clustered_rows(orc, n, n_centres=32)
    base = orc.f16_bits(x)
    adj, degs = knn_graph(x, deg, rng)
    vecs = mse.VectorList.from_f16s(base, D)
    searcher = mse.Searcher(vecs)
    dgraph = mse.DeviceGraph(mse.IndexGraph(adj, degs))
    mse.set_entries(dgraph, vecs, np.sort(rng.choice(n, 64, replace=False)).astype(np.uint32))
    q32 = clustered_rows(orc, nq, n_centres=32, seed=5).astype(np.float32)
    plain = mse.disk_query_topk(searcher, None, None, dgraph, q32, k, None, None, None, True, beam, L)
    assert searcher.beam_timing(2)["launches"] == 0                     # nothing was measured while the hook was off
    got = mse.disk_query_topk(searcher, None, None, dgraph, q32, k, None, None, None, True, beam, L)
    m = searcher.beam_timing(0)
    assert np.array_equal(got[0], plain[0]) and np.array_equal(got[1], plain[1])
    for key in ("cmps", "n_visited", "pq_cmps"):
        assert np.array_equal(got[2][key], plain[2][key])
    assert m["launches"] >= 1 and m["queries"] == nq and m["kernel_ms"] > 0
    assert m["nodes_fetched"] == int(got[2]["cmps"].sum())
    assert m["adc_scored"] == 0 and m["rows_scored"] > m["nodes_fetched"]
    assert m["rows_scored"] <= nq + m["nodes_fetched"] * deg                              # the entry point + at most every neighbour of every fetched node
    assert m["iterations"] * beam >= m["nodes_fetched"] and m["iterations_replayed"] <= m["iterations"]
    after = mse.disk_query_topk(searcher, None, None, dgraph, q32, k, None, None, None, True, beam, L)
    assert searcher.beam_timing(0)["launches"] == m["launches"]        # off again: the totals stand still
    assert np.array_equal(after[0], plain[0])
    # the reference's default: neighbours scored by ADC (random codec: the counters are what is looked at)
    cents = (rng.standard_normal((256, D)) / np.sqrt(D)).astype(np.float32)
    T = np.linalg.qr(rng.standard_normal((D, D)))[0].astype(np.float32)
    pq = mse.ProductQuantizer(cents, T, 18, D)
    codes = mse.Codes(pq.quantize_batch(x.astype(np.float32)), None)
    plain = mse.disk_query_topk(searcher, pq, codes, dgraph, q32, k, None, None, None, False, beam, L)
    searcher.beam_timing(2)
    got = mse.disk_query_topk(searcher, pq, codes, dgraph, q32, k, None, None, None, False, beam, L)
    m = searcher.beam_timing(0)
    assert np.array_equal(got[0], plain[0]) and np.array_equal(got[1], plain[1]) and np.array_equal(got[2]["pq_cmps"], plain[2]["pq_cmps"])
    assert m["nodes_fetched"] == int(got[2]["cmps"].sum()) and m["rows_scored"] == m["nodes_fetched"]
    assert 0 < m["adc_scored"] <= int(got[2]["pq_cmps"].sum())          # pq_cmps also counts the re-offers of the pre-buffer quirk


def test_entry_step_small_and_large_batches_agree(gpu, mse, orc):
    """The row-table entry step has two forms -- two launches of exact dots for a small batch, the brute-force searcher's matrix-core
    path for a large one (nq x entries beyond 2^22) -- and both are the exact top-1 with ties to the lower row: a large batch, the
    same queries in small calls, and the oracle's top-1 give the same start nodes (duplicate entry rows make ties certain)."""
    rng = np.random.default_rng(43)
    n, deg, E, nq, k, L = 9000, 10, 8000, 600, 5, 16
    x = clustered_rows(orc, n, n_centres=48)
    x[4000:4100] = x[100:200]                                          # duplicate rows: equal scores for every query
    base = orc.f16_bits(x)
    adj, degs = knn_graph(x, deg, rng)
    vecs = mse.VectorList.from_f16s(base, D)
    searcher = mse.Searcher(vecs)
    dgraph = mse.DeviceGraph(mse.IndexGraph(adj, degs))
    entry_ids = np.sort(rng.choice(n, E, replace=False)).astype(np.uint32)
    mse.set_entries(dgraph, vecs, entry_ids)
    qh = orc.f16_bits(clustered_rows(orc, nq, n_centres=48, seed=320))
    qh[:20] = base[100:120]                                            # queries whose best entries are duplicated rows
    _, best = orc.bruteforce_topk(base[entry_ids], qh, 1)
    want = mse.disk_query_topk(searcher, None, None, dgraph, qh, k, entry_ids[best[:, 0]], None, None, True, 2, L)
    large = mse.disk_query_topk(searcher, None, None, dgraph, qh, k, None, None, None, True, 2, L)           # 600 x 8000 > 2^22
    assert np.array_equal(large[0], want[0]) and np.array_equal(large[1], want[1])
    for q0 in (0, 17, 200):
        small = mse.disk_query_topk(searcher, None, None, dgraph, qh[q0:q0 + 33], k, None, None, None, True, 2, L)
        assert np.array_equal(small[0], want[0][q0:q0 + 33]) and np.array_equal(small[1], want[1][q0:q0 + 33])


@pytest.mark.parametrize("disable_pq", [True, False])
def test_request_path_with_the_handlers_deduplication(gpu, mse, orc, disable_pq):
    """The handler drops visited records that resemble an already kept one (dot > 0.95 of the f32-widened vectors, visit order)
    BEFORE it orders them (src/query_disk_index.rs:482-527).  mse_graph_set_dedup runs that inside mse_disk_query_topk for every
    query of the batch: against the oracle's search + orc.dedup_keep + sort, on a base in which a fifth of the rows have a
    near-duplicate; off again, the plain k best visited records come back."""
    rng = np.random.default_rng(61)
    n, deg, k, L, nq = 3000, 14, 10, 40, 29
    x = clustered_rows(orc, n, n_centres=32)
    twins = rng.choice(n // 2, 600, replace=False)
    x[n // 2 + np.arange(600)] = x[twins] + rng.standard_normal((600, D)).astype(np.float32) * np.float32(0.002)   # dot ~ 0.995 with its twin
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    base = orc.f16_bits(x)
    cents, T = train_pq(orc, x[:2000], iters=2)
    opq, gpq = orc.PQ(cents, T, 18, D), mse.ProductQuantizer(cents, T, 18, D)
    codes = opq.quantize_batch(orc.f16_to_f32(base))
    desc = np.zeros((n, 4), np.uint8)
    adj, degs = knn_graph(x, deg, rng)
    vecs = mse.VectorList.from_f16s(base, D)
    searcher = mse.Searcher(vecs)
    gcodes = mse.Codes(codes, desc)
    dgraph = mse.DeviceGraph(mse.IndexGraph(adj, degs))
    qs = clustered_rows(orc, nq, n_centres=32, seed=330).astype(np.float32)
    qs[:5] = x[twins[:5]]                                              # queries whose best matches are twin pairs
    qh = orc.f16_bits(qs)
    starts = rng.integers(0, n, size=nq).astype(np.uint32)
    starts[:5] = twins[:5]
    luts = np.stack([opq.preprocess_query(q) for q in qs])
    args = (searcher, gpq, gcodes, dgraph, qh, k, starts, luts, None, disable_pq, 4, L)
    plain = mse.disk_query_topk(*args)
    mse.set_dedup(dgraph, mse.DUPLICATES_THRESHOLD)
    got = mse.disk_query_topk(*args)
    changed = 0
    for i in range(nq):
        _, ovids, ovsc, _, _ = orc.disk_greedy_search(base, adj, degs, codes, desc, int(starts[i]), qh[i], luts[i], None, disable_pq, 4, L, None)
        keep = orc.dedup_keep(base[ovids], 0.95).astype(bool)
        kid, ksc = ovids[keep], ovsc[keep]
        order = sorted(range(len(kid)), key=lambda j: (-int(ksc[j]), int(kid[j])))[:k]
        want_ids = np.full(k, 0xFFFFFFFF, np.uint32)
        want_sc = np.full(k, np.iinfo(np.int64).min, np.int64)
        want_ids[:len(order)] = kid[order]
        want_sc[:len(order)] = ksc[order]
        assert np.array_equal(got[0][i], want_ids) and np.array_equal(got[1][i], want_sc), i
        assert int(got[2]["n_visited"][i]) == len(ovids)
        changed += not np.array_equal(got[0][i], plain[0][i])
    assert changed >= 5                                                # the filter really removed something
    # one query per call through the coalescer: the same
    one = mse.disk_query_topk(searcher, gpq, gcodes, dgraph, qh[3:4], k, starts[3:4], luts[3:4], None, disable_pq, 4, L)
    assert np.array_equal(one[0][0], got[0][3]) and np.array_equal(one[1][0], got[1][3])
    mse.set_dedup(dgraph, 0.0)
    again = mse.disk_query_topk(*args)
    assert np.array_equal(again[0], plain[0]) and np.array_equal(again[1], plain[1])
