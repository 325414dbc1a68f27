"""Parity of the HIP brute-force path with the CPU oracle, through the C ABI (mse package).
Bit-exact bar: i64 scores and returned indices."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, SEED_BASE, SEED_QUERY

pytestmark = pytest.mark.gpu
D = 1152


def h(x):
    return np.asarray(x, np.float16).view(np.uint16)


def test_native_library_is_loaded(gpu, mse):
    from mse import ffi
    assert os.path.exists(ffi.LIB_PATH)
    maps = open("/proc/self/maps").read()
    assert "libmse_hip.so" in maps
    assert b"gfx950" in ffi.lib().mse_version()


def test_generator_is_bit_identical_to_oracle(gpu, mse, orc):
    for d, n, first in ((1152, 1000, 0), (1152, 64, 123456789012), (64, 300, 5), (4096, 10, 0)):
        vl = mse.VectorList.generate(SEED_BASE, first, n, d)
        assert np.array_equal(vl.rows(0, n), orc.gen_rows_f16(SEED_BASE, first, n, d))


def test_fast_dot_known_answers(gpu, mse, orc):
    ones = np.full(D, 0x3C00, np.uint16)
    assert mse.fast_dot_noprefetch(ones, ones) == D << 32
    y = h(np.arange(1, D + 1) / 1024.0)
    for e in (0, 7, 8, 15, 16, 24, 31, 32, 63, 777, D - 1):
        x = np.zeros(D, np.uint16)
        x[e] = h(2.0)
        assert mse.fast_dot(x, y, y) == orc.fast_dot(x, y)
    rng = np.random.default_rng(1)
    # arbitrary bit patterns incl. f16 subnormals, large magnitudes (no inf/nan), both signs
    x = rng.integers(0, 0x7BFF, size=(40, D), dtype=np.uint16) | (rng.integers(0, 2, size=(40, D), dtype=np.uint16) << 15)
    for i in range(0, 40, 2):
        assert mse.fast_dot_noprefetch(x[i], x[i + 1]) == orc.fast_dot(x[i], x[i + 1])
    # reduction tree order: one live step so the 32 products are the 32 accumulator slots
    for seed in range(10):
        r = np.random.default_rng(seed)
        a = np.zeros(64, np.float16)
        b = np.zeros(64, np.float16)
        a[:32] = (2.0 ** r.integers(-10, 14, 32)) * r.uniform(1.0, 2.0, 32) * r.choice([1.0, -1.0], 32)
        b[:32] = 1.0
        assert mse.fast_dot(h(a), h(b)) == orc.fast_dot(h(a), h(b))
    # saturation / NaN of `as i64`
    big = np.full(64, h(60000.0), np.uint16)
    assert mse.fast_dot(big, big) == orc.fast_dot(big, big) == (1 << 63) - 1
    nan = big.copy()
    nan[3] = 0x7E00
    assert mse.fast_dot(nan, big) == orc.fast_dot(nan, big) == 0


def test_golden_fixture(gpu, mse):
    g = np.load(os.path.join(GOLDEN, "bruteforce_256x1152.npz"))
    vl = mse.VectorList.generate(int(g["seed_base"]), 0, 256)
    assert np.array_equal(vl.rows(0, 8), g["base_head"])
    s = mse.Searcher(vl)
    for i in range(8):
        assert np.array_equal(s.scores(g["queries"][i]), g["scores"][i])
    for mode in (mse.MODE_EXACT, mse.MODE_MFMA, mse.MODE_AUTO):
        sc, ids = s.bruteforce_topk(g["queries"], 10, mode)
        assert np.array_equal(sc, g["top_scores"]) and np.array_equal(ids, g["top_ids"])


@pytest.mark.parametrize("n,d", [(1, 1152), (63, 1152), (64, 1152), (65, 1152), (1000, 64), (1000, 1024),
                                 (4097, 1152), (20000, 1152), (70000, 256)])
def test_scores_match_oracle(gpu, mse, orc, n, d):
    base = orc.gen_rows_f16(SEED_BASE, 0, n, d)
    q = orc.gen_rows_f16(SEED_QUERY, 0, 3, d)
    vl = mse.VectorList.from_f16s(base, d)
    s = mse.Searcher(vl)
    for i in range(3):
        assert np.array_equal(s.scores(q[i]), orc.score_all(base, q[i]))
    ids = np.random.default_rng(0).integers(0, n, 100).astype(np.uint32)
    assert np.array_equal(s.score_rows(ids, q[0]), orc.score_rows(base, ids, q[0]))
    r = s.ranks(q[1], ids[:20])
    assert np.array_equal(r, orc.ranks_from_scores(orc.score_all(base, q[1]))[ids[:20]])


@pytest.mark.parametrize("mode", ["exact", "mfma"])
@pytest.mark.parametrize("n,nq,k", [(1, 1, 1), (5, 3, 10), (31, 2, 10), (33, 9, 5), (1000, 17, 10), (5000, 1, 1000),
                                    (20000, 130, 10), (40000, 8, 100), (30000, 256, 10), (9000, 300, 7), (15000, 192, 10), (3333, 160, 3), (21000, 320, 10), (5000, 330, 4), (7000, 700, 5)])
def test_topk_matches_oracle(gpu, mse, orc, mode, n, nq, k):
    base = orc.gen_rows_f16(SEED_BASE, 0, n)
    q = orc.gen_rows_f16(SEED_QUERY, 0, nq)
    s = mse.Searcher(mse.VectorList.from_f16s(base, D))
    m = mse.MODE_EXACT if mode == "exact" else mse.MODE_MFMA
    sc, ids = s.bruteforce_topk(q, k, m)
    ws, wi = orc.bruteforce_topk(base, q, k)
    assert np.array_equal(ids, wi)
    assert np.array_equal(sc, ws)


@pytest.mark.parametrize("d,nq", [(192, 300), (64, 321), (128, 320), (256, 577), (448, 290)])
def test_wide_passes_at_other_widths(gpu, mse, orc, d, nq):
    # the 320-query pass needs an even number of 64-element K blocks; odd widths stay on 256-query passes
    assert mse.ffi.lib().mse_queries_per_pass_max(d) == (320 if (d // 64) % 2 == 0 else 256)
    base = orc.gen_rows_f16(SEED_BASE, 0, 6000, d)
    q = orc.gen_rows_f16(SEED_QUERY, 0, nq, d)
    s = mse.Searcher(mse.VectorList.from_f16s(base, d))
    sc, ids = s.bruteforce_topk(q, 6, mse.MODE_MFMA)
    ws, wi = orc.bruteforce_topk(base, q, 6)
    assert np.array_equal(ids, wi) and np.array_equal(sc, ws)


@pytest.mark.parametrize("n,nq,k", [(300, 2000, 1), (4096, 1500, 3), (33, 700, 5)])
def test_many_queries_against_a_small_base(gpu, mse, orc, n, nq, k):
    # a small base takes all its queries in ONE matrix-core call: the full passes side by side in one launch, the remainder after
    # them, one tournament / re-score / certificate over all queries (api.hip mfma_call_tile; the request path's entry step)
    base = orc.gen_rows_f16(SEED_BASE, 0, n)
    q = orc.gen_rows_f16(SEED_QUERY, 0, nq)
    s = mse.Searcher(mse.VectorList.from_f16s(base, D))
    sc, ids = s.bruteforce_topk(q, k, mse.MODE_MFMA)
    ws, wi = orc.bruteforce_topk(base, q, k)
    assert np.array_equal(ids, wi) and np.array_equal(sc, ws)


def test_ties_break_by_lower_id(gpu, mse, orc):
    # duplicated rows => exactly equal scores; order must be (score desc, id asc) in every mode
    rows = orc.gen_rows_f16(SEED_BASE, 0, 40)
    base = np.concatenate([rows] * 30)            # 1200 rows, each distinct row appears 30 times
    q = orc.gen_rows_f16(SEED_QUERY, 0, 20)
    s = mse.Searcher(mse.VectorList.from_f16s(base, D))
    ws, wi = orc.bruteforce_topk(base, q, 64)
    for mode in (mse.MODE_EXACT, mse.MODE_MFMA):
        sc, ids = s.bruteforce_topk(q, 64, mode)
        assert np.array_equal(ids, wi) and np.array_equal(sc, ws)
    assert np.all(wi[:, 1] == wi[:, 0] + 40)      # the tie really is resolved by id


def test_unnormalised_and_adversarial_data(gpu, mse, orc):
    # large norms, sorted-by-score layouts (worst case for threshold schemes), constant rows
    rng = np.random.default_rng(5)
    q = orc.f16_bits((rng.standard_normal((12, D)) * 3).astype(np.float32))
    base_f = (rng.standard_normal((6000, D)) * rng.uniform(0.01, 20, (6000, 1))).astype(np.float32)
    order = np.argsort(base_f @ orc.f16_to_f32(q[0]))          # ascending: best rows arrive last
    base = orc.f16_bits(base_f[order])
    base[100:200] = base[100]                                   # a run of identical rows
    s = mse.Searcher(mse.VectorList.from_f16s(base, D))
    ws, wi = orc.bruteforce_topk(base, q, 10)
    for mode in (mse.MODE_EXACT, mse.MODE_MFMA):
        sc, ids = s.bruteforce_topk(q, 10, mode)
        assert np.array_equal(ids, wi) and np.array_equal(sc, ws)


def test_empty_and_degenerate_inputs(gpu, mse, orc):
    vl = mse.VectorList.from_f16s(np.zeros((0, D), np.uint16), D)
    s = mse.Searcher(vl)
    q = orc.gen_rows_f16(SEED_QUERY, 0, 2)
    sc, ids = s.bruteforce_topk(q, 3)
    assert np.all(ids == 0xFFFFFFFF) and np.all(sc == -(1 << 63))
    base = orc.gen_rows_f16(SEED_BASE, 0, 10)
    s = mse.Searcher(mse.VectorList.from_f16s(base, D))
    sc, ids = s.bruteforce_topk(q[:0], 3)
    assert sc.shape == (0, 3)
    assert list(s.score_rows(np.array([3, 99, 0xFFFFFFFF], np.uint32), q[0])[1:]) == [-(1 << 63)] * 2
    with pytest.raises(mse.MseError):
        s.bruteforce_topk(q, 5000)


def test_certificate_widens_on_near_ties(gpu, mse, orc):
    # many rows whose scores differ by less than the certified error bound: the MFMA stage cannot
    # separate them, the certificate must fail and the search must widen (or fall back) and still
    # return the exact answer
    rng = np.random.default_rng(7)
    q = orc.gen_rows_f16(SEED_QUERY, 0, 16)
    proto = orc.f16_to_f32(orc.gen_rows_f16(SEED_BASE, 0, 1)[0])
    base_f = np.tile(proto, (3000, 1))
    flip = rng.integers(0, D, 3000)
    base_f[np.arange(3000), flip] *= (1.0 + 2.0 ** -9)         # one-ulp-ish perturbations of one row
    base = orc.f16_bits(base_f)
    s = mse.Searcher(mse.VectorList.from_f16s(base, D))
    ws, wi = orc.bruteforce_topk(base, q, 10)
    sc, ids = s.bruteforce_topk(q, 10, mse.MODE_MFMA)
    assert np.array_equal(ids, wi) and np.array_equal(sc, ws)
    st = s.last_stats()
    assert st["widened_queries"] > 0 and st["max_groups"] > 18


def test_sharded_equals_whole(gpu, mse, orc):
    # the multi-GPU decomposition on one device: two shards with id offsets + device merge
    import torch
    from mse import shard
    n, nq, k = 30000, 20, 10
    q = orc.gen_rows_f16(SEED_QUERY, 0, nq)
    whole = mse.Searcher(mse.VectorList.generate(SEED_BASE, 0, n))
    ws, wi = whole.bruteforce_topk(q, k, mse.MODE_MFMA)
    qd = torch.from_numpy(q.view(np.int16)).cuda()
    gs = torch.empty((2, nq, k), dtype=torch.int64, device="cuda")
    gi = torch.empty((2, nq, k), dtype=torch.int32, device="cuda")
    searchers = []
    for r in range(2):
        lo, hi = shard.shard_range(n, r, 2)
        sr = mse.Searcher(mse.VectorList.generate(SEED_BASE, lo, hi - lo))
        sr.bruteforce_topk_dev(qd.data_ptr(), nq, k, gs[r].data_ptr(), gi[r].data_ptr(), mse.MODE_MFMA, id_offset=lo)
        searchers.append(sr)
    from mse import ffi
    ffi.check(ffi.lib().mse_device_synchronize())
    out_s = torch.empty((nq, k), dtype=torch.int64, device="cuda")
    out_i = torch.empty((nq, k), dtype=torch.int32, device="cuda")
    searchers[0].merge_topk_dev(gs.data_ptr(), gi.data_ptr(), 2, nq, k, out_s.data_ptr(), out_i.data_ptr())
    ffi.check(ffi.lib().mse_device_synchronize())
    assert np.array_equal(out_s.cpu().numpy(), ws)
    assert np.array_equal(out_i.cpu().numpy().view(np.uint32), wi)
    # and the torch-side merge used by the gloo test agrees with the device merge
    ms, mi = shard.merge_topk_torch(gs.permute(1, 0, 2).reshape(nq, 2 * k), gi.permute(1, 0, 2).reshape(nq, 2 * k), k)
    assert np.array_equal(ms.cpu().numpy(), ws) and np.array_equal(mi.cpu().numpy().astype(np.uint32), wi)


def test_full_size_properties_1e7(gpu, mse, orc):
    """BASELINE config 3 size (1e7 x 1152): size-independent properties instead of an oracle scan."""
    n, nq, k = 10_000_000, 136, 10
    vl = mse.VectorList.generate(SEED_BASE, 0, n)
    s = mse.Searcher(vl)
    q = orc.gen_rows_f16(SEED_QUERY, 0, nq)
    # plant known winners: queries 0..3 are exact copies of rows deep inside the base
    planted = [0, 4_999_999, 9_999_999, 1234567]
    for j, r in enumerate(planted):
        q[j] = orc.gen_rows_f16(SEED_BASE, r, 1)[0]
    sm, im = s.bruteforce_topk(q, k, mse.MODE_MFMA)
    se, ie = s.bruteforce_topk(q[:8], k, mse.MODE_EXACT)
    assert np.array_equal(im[:8], ie) and np.array_equal(sm[:8], se)           # two independent kernels agree
    assert [int(im[j, 0]) for j in range(4)] == planted                        # a row is its own best match
    assert np.all(sm[:, :-1] >= sm[:, 1:])                                     # sorted
    assert np.all(im < n) and all(len(set(r)) == k for r in im.tolist())       # valid, distinct
    # returned scores are the oracle's scores of the returned rows (rows regenerated on the host)
    for qi in (0, 5, 77, 135):
        for j in (0, 3, 9):
            row = orc.gen_rows_f16(SEED_BASE, int(im[qi, j]), 1)[0]
            assert orc.fast_dot(q[qi], row) == int(sm[qi, j])
    # k-th score bounds every non-returned row: check on a random sample of rows with the oracle
    rng = np.random.default_rng(3)
    sample = rng.integers(0, n, 300)
    rows = np.stack([orc.gen_rows_f16(SEED_BASE, int(r), 1)[0] for r in sample])
    for qi in (5, 77):
        sc = orc.score_all(rows, q[qi])
        outside = ~np.isin(sample, im[qi])
        assert np.all(sc[outside] <= sm[qi, -1])
    # idempotence
    sm2, im2 = s.bruteforce_topk(q, k, mse.MODE_MFMA)
    assert np.array_equal(sm, sm2) and np.array_equal(im, im2)
    assert s.last_stats()["widened_queries"] == 0


def test_mfma_error_bound_is_measured_not_assumed(gpu, mse, orc):
    """The exactness certificate of the batched scan rests on |MFMA score - exact-order score| <= eps * |q| * |x| with
    eps = 2.8e-4 (api.hip mfma_pass / DESIGN 3.1).  Measure the left side: a 1e7-row base in which every row is repeated 32
    times, so that the scan's per-32-row maximum IS the row's MFMA score, against the exact-order scores of the 312 500
    distinct rows, for 128 and 256 queries (40 M and 80 M (row, query) pairs).  The bound must hold with a margin of 4."""
    import ctypes as C
    import torch
    from mse import ffi
    n_distinct = 312_500
    rows = orc.gen_rows_f16(SEED_BASE, 0, n_distinct)
    small = mse.Searcher(mse.VectorList.from_f16s(rows, D))
    rep = torch.from_numpy(rows.view(np.int16)).cuda().repeat_interleave(32, dim=0).contiguous()     # 1e7 x 1152, 23 GB
    torch.cuda.synchronize()
    big = mse.Searcher(mse.VectorList.wrap_device(rep.data_ptr(), n_distinct * 32, D, keepalive=rep))
    row_norm = np.linalg.norm(orc.f16_to_f32(rows).astype(np.float64), axis=1)
    worst = 0.0
    for nq in (128, 256):
        q = orc.gen_rows_f16(SEED_QUERY, 1000, nq)
        q[: nq // 2] = (orc.f16_to_f32(q[: nq // 2]) * np.float32(3.0)).astype(np.float16).view(np.uint16)   # queries need not be unit norm
        got = np.empty((n_distinct, nq), np.float32)
        ffi.check(ffi.lib().mse_debug_mfma_group_max(big._h, q.ctypes.data_as(ffi.u16p), nq, got.ctypes.data_as(ffi.f32p)))
        q_norm = np.linalg.norm(orc.f16_to_f32(q).astype(np.float64), axis=1)
        for j in range(nq):
            exact = small.scores(q[j]).astype(np.float64) / 4294967296.0
            ratio = np.abs(got[:, j].astype(np.float64) - exact) / (row_norm * q_norm[j])
            worst = max(worst, float(ratio.max()))
    print("max |mfma - exact| / (|q||x|) =", worst)
    assert worst > 0                                  # the two orders do differ: the certificate is not vacuous
    assert 2.8e-4 >= 4 * worst, worst


def test_developer_knobs_do_nothing_in_the_product_build(gpu, mse, orc, monkeypatch):
    """Every environment variable that used to select a timing ablation (answers wrong by design) or a superseded kernel is set
    to its most destructive value; the product library never reads them, so the batched answer still equals the oracle's and the
    exact-order kernel's, and a small PQ scan / dedup still equal theirs."""
    for name, val in [("MSE_SCAN_ABL", "1"), ("MSE_SCAN_2D", "162"), ("MSE_SCAN_S", "1"), ("MSE_ATT_ABL", "1"), ("MSE_ATT64_ABL", "2"),
                      ("MSE_ATT_WAVES", "4"), ("MSE_ATT_QT", "4"), ("MSE_ATT_TILE32", "1"), ("MSE_GEMM_RANDOM", "1"),
                      ("MSE_GEMM_OLD256", "1"), ("MSE_GEMM_128", "1"), ("MSE_GEMM_NOPERSIST", "1"), ("MSE_GEMM_STAGGER", "7"),
                      ("MSE_GEMM_NONARROW", "1"), ("MSE_PQ_OLDTRANSFORM", "1"), ("MSE_PQ_OLDQUANT", "1"), ("MSE_PQ_OLDSCAN", "1"),
                      ("MSE_DEDUP_OLD", "1")]:
        monkeypatch.setenv(name, val)
    n, nq, k = 30_000, 256, 10
    rows = orc.gen_rows_f16(SEED_BASE, 0, n)
    q = orc.gen_rows_f16(SEED_QUERY, 0, nq)
    searcher = mse.Searcher(mse.VectorList.generate(SEED_BASE, 0, n))
    s, i = searcher.bruteforce_topk(q, k, mse.MODE_MFMA)          # the 256-query pass: the kernel the ablations lived in
    ws, wi = orc.bruteforce_topk(rows, q, k)
    assert np.array_equal(i, wi) and np.array_equal(s, ws)
    se, ie = searcher.bruteforce_topk(q[:8], k, mse.MODE_EXACT)
    assert np.array_equal(ie, wi[:8]) and np.array_equal(se, ws[:8])


def test_only_the_queries_that_need_it_are_widened(gpu, mse, orc):
    """Near-duplicates around ONE query's k-th score (600 copies of its best row, spread over ~600 32-row groups, far more than the
    k + 8 groups the first round nominates): that query's certificate fails and it alone is carried on through wider rounds (its
    ten results are the ten lowest ids among the copies, exact ties), the other 199 queries are certified in the first round.
    Results equal the oracle's in both modes; a second planted query whose copies sit in 2200 groups -- more than the widest nomination
    (2048) -- ends in the exact scan and is still right."""
    n, nq, k = 100000, 200, 10
    base = orc.gen_rows_f16(SEED_BASE, 21, n).copy()
    q = orc.gen_rows_f16(SEED_QUERY, 21, nq)
    rng = np.random.default_rng(8)
    groups = rng.permutation(n // 32)
    pos = np.sort(groups[:600] * 32 + rng.integers(0, 32, 600))
    base[pos] = q[3]                                           # 600 exact copies of query 3 itself, one per 32-row group
    pos2 = np.sort(groups[600:600 + 2200] * 32 + rng.integers(0, 32, 2200))
    base[pos2] = q[77]                                         # one copy in each of 2200 groups: more than any widening reaches (2048)
    ws, wi = orc.bruteforce_topk(base, q, k)
    assert np.array_equal(wi[3], np.sort(pos)[:k].astype(np.uint32)) and np.array_equal(wi[77], np.sort(pos2)[:k].astype(np.uint32))
    s = mse.Searcher(mse.VectorList.from_f16s(base, D))
    sc, ids = s.bruteforce_topk(q, k, mse.MODE_MFMA)
    assert np.array_equal(ids, wi) and np.array_equal(sc, ws)
    st = s.last_stats()
    assert st["widened_queries"] == 2, st                      # queries 3 and 77, nobody else
    sc, ids = s.bruteforce_topk(q[:8], k, mse.MODE_EXACT)
    assert np.array_equal(ids, wi[:8]) and np.array_equal(sc, ws[:8])
