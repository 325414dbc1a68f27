"""Pins the CPU oracle (oracle/mse_oracle.c) with hand-derivable known answers and the committed
fixtures.  The reference ships no golden vectors for this path, so these known
answers are derived by hand from the reference's source (file:line in each test)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, SEED_BASE, SEED_QUERY

ONE = 0x3C00  # f16 1.0
D = 1152


def h(x):
    return np.asarray(x, np.float16).view(np.uint16)


def test_built_with_reference_intrinsics(orc):
    # the reference's kernel is AVX2+F16C+FMA (diskann/src/vector.rs:192-306); so is the oracle here
    assert orc.lib().orc_have_avx2() == 1


def test_scale_dot_result_rust_as_cast(orc):
    # vector.rs:408-416: `(x * SCALE) as i64` -- truncation toward zero, saturation, NaN -> 0
    L = orc.lib()
    assert L.orc_scale_dot_result(1.0) == 1 << 32
    assert L.orc_scale_dot_result(-1.0) == -(1 << 32)
    assert L.orc_scale_dot_result(0.5) == 1 << 31
    assert L.orc_scale_dot_result(np.float32(1e-10)) == 0                 # 0.4294 truncates to 0
    assert L.orc_scale_dot_result(np.float32(-3e-10)) == -1               # -1.288 truncates toward zero
    assert L.orc_scale_dot_result(np.float32(1e30)) == (1 << 63) - 1      # saturates
    assert L.orc_scale_dot_result(np.float32(-1e30)) == -(1 << 63)
    assert L.orc_scale_dot_result(float("nan")) == 0
    assert L.orc_scale_dot_result(float("inf")) == (1 << 63) - 1
    assert L.orc_scale_dot_result_f64(0.25) == 1 << 30
    assert L.orc_scale_dot_result_f64(float("nan")) == 0


def test_f16_conversions_match_ieee(orc):
    allh = np.arange(65536, dtype=np.uint16)
    a = orc.f16_to_f32(allh)
    b = allh.view(np.float16).astype(np.float32)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    rng = np.random.default_rng(0)
    f = (rng.standard_normal(100000) * 10.0 ** rng.integers(-9, 5, 100000)).astype(np.float32)
    f = np.concatenate([f, np.array([0.0, -0.0, 65504.0, 65519.9, 65520.0, 1e-8, 5.96e-8, 2.98e-8, 2.99e-8], np.float32)])
    with np.errstate(over="ignore"):
        assert np.array_equal(orc.f16_bits(f), f.astype(np.float16).view(np.uint16))


def test_fast_dot_all_ones(orc):
    ones = np.full(D, ONE, np.uint16)
    assert orc.fast_dot(ones, ones) == D << 32
    assert orc.fast_dot_scalar(ones, ones) == D << 32


def test_fast_dot_one_hot_hits_every_accumulator_slot(orc):
    # element e lands in accumulator (e mod 32)/8, lane e mod 8 (vector.rs:279-291); a one-hot
    # product must come through the whole reduction tree untouched
    y = h(np.arange(1, D + 1) / 1024.0)
    for e in list(range(0, 64)) + [D - 1, D - 32, 777]:
        x = np.zeros(D, np.uint16)
        x[e] = h(2.0)
        want = int(np.float32(2.0) * np.float32(np.float16((e + 1) / 1024.0)) * np.float32(2 ** 32))
        assert orc.fast_dot(x, y) == want == orc.fast_dot_scalar(x, y)


def test_fast_dot_reduction_tree_order(orc):
    # One non-zero step (n = 64, second step zero): the 32 products ARE the 32 accumulator slots.
    # Expected value computed literally from the reduction at vector.rs:295-303; a different
    # association (plain left-to-right) must disagree for at least some inputs, or the test is blind.
    f = np.float32
    differs = 0
    for seed in range(40):
        rng = np.random.default_rng(seed)
        mags = (2.0 ** rng.integers(-10, 14, 32)) * rng.uniform(1.0, 2.0, 32) * rng.choice([1.0, -1.0], 32)
        x = np.zeros(64, np.float16)
        x[:32] = mags
        y = np.zeros(64, np.float16)
        y[:32] = 1.0
        acc = x[:32].astype(np.float32)
        a = [f(acc[l] + acc[8 + l]) for l in range(8)]
        b = [f(acc[16 + l] + acc[24 + l]) for l in range(8)]
        hh = [f(a[0] + a[1]), f(a[2] + a[3]), f(b[0] + b[1]), f(b[2] + b[3]),
              f(a[4] + a[5]), f(a[6] + a[7]), f(b[4] + b[5]), f(b[6] + b[7])]
        s = [f(hh[0] + hh[4]), f(hh[1] + hh[5]), f(hh[2] + hh[6]), f(hh[3] + hh[7])]
        want = f(f(f(s[0] + s[1]) + s[2]) + s[3])
        assert np.float32(orc.fast_dot_f32(h(x), h(y))) == want
        naive = f(0)
        for v in acc:
            naive = f(naive + v)
        differs += int(naive != want)
    assert differs > 5


def test_fast_dot_uses_fused_multiply_add(orc):
    # acc = fma(x, y, acc): with x*y needing more than 24 bits relative to acc the fused and the
    # unfused results differ (vector.rs:288 `_mm256_fmadd_ps`)
    x = np.zeros(64, np.float16)
    y = np.zeros(64, np.float16)
    x[0], y[0] = 1.0, 1.0               # step 0: acc slot0 = 1
    x[32], y[32] = np.float16(2.0 ** -12 + 2.0 ** -22 * 0), np.float16(2.0 ** -12)
    x[32] = np.float16(1.0009765625 * 2.0 ** -12)   # (1+2^-10) * 2^-12
    y[32] = np.float16(1.0009765625 * 2.0 ** -12)
    prod_exact = (1.0009765625 * 2.0 ** -12) ** 2
    want = np.float32(1.0 + prod_exact)              # single rounding
    assert np.float32(orc.fast_dot_f32(h(x), h(y))) == want


def test_avx_and_scalar_restatements_agree(orc):
    base = orc.gen_rows_f16(SEED_BASE, 0, 128)
    q = orc.gen_rows_f16(SEED_QUERY, 0, 4)
    for i in range(128):
        for j in range(4):
            assert orc.fast_dot(base[i], q[j]) == orc.fast_dot_scalar(base[i], q[j])
    # denormal halves and large values
    rng = np.random.default_rng(1)
    x = rng.integers(0, 0x7BFF, size=(16, 1152), dtype=np.uint16) | (rng.integers(0, 2, size=(16, 1152), dtype=np.uint16) << 15)
    for i in range(0, 16, 2):
        assert orc.fast_dot(x[i], x[i + 1]) == orc.fast_dot_scalar(x[i], x[i + 1])


def test_generator_is_unit_norm_and_deterministic(orc):
    a = orc.gen_rows_f16(SEED_BASE, 5, 3)
    b = orc.gen_rows_f16(SEED_BASE, 0, 8)[5:8]
    assert np.array_equal(a, b)
    n = np.linalg.norm(orc.f16_to_f32(a).astype(np.float64), axis=1)
    assert np.all(np.abs(n - 1.0) < 5e-4)
    ints = orc.gen_row_ints(SEED_BASE, 0)
    assert ints.min() >= -131070 and ints.max() <= 131070 and abs(float(ints.mean())) < 5000


def test_golden_bruteforce(orc):
    g = np.load(os.path.join(GOLDEN, "bruteforce_256x1152.npz"))
    base = orc.gen_rows_f16(int(g["seed_base"]), 0, 256)
    assert np.array_equal(base[:8], g["base_head"])
    q = orc.gen_rows_f16(int(g["seed_query"]), 0, 8)
    assert np.array_equal(q, g["queries"])
    for i in range(8):
        assert np.array_equal(orc.score_all(base, q[i]), g["scores"][i])
    s, ids = orc.bruteforce_topk(base, q, 10)
    assert np.array_equal(s, g["top_scores"]) and np.array_equal(ids, g["top_ids"])
    # top-k really is the sorted head of the scores, ties by id
    for i in range(8):
        order = np.lexsort((np.arange(256), -g["scores"][i]))[:10]
        assert np.array_equal(order.astype(np.uint32), ids[i])


def test_topk_tie_break_and_padding(orc):
    scores = np.array([5, 9, 9, 1, 9, -3], np.int64)
    s, ids = orc.topk_from_scores(scores, 4)
    assert list(ids) == [1, 2, 4, 0] and list(s) == [9, 9, 9, 5]
    base = orc.gen_rows_f16(SEED_BASE, 0, 3)
    q = orc.gen_rows_f16(SEED_QUERY, 0, 1)
    s, ids = orc.bruteforce_topk(base, q, 5)
    assert list(ids[0][3:]) == [0xFFFFFFFF] * 2 and list(s[0][3:]) == [-(1 << 63)] * 2


def test_ranks_and_recall(orc):
    # src/query_disk_index.rs:271-273,321-326
    scores = np.array([10, 50, 50, 7, 99], np.int64)
    r = orc.ranks_from_scores(scores)
    assert list(r) == [3, 1, 2, 4, 0]
    assert orc.recall_at_k(r, np.array([4, 1, 3], np.uint32), 3) == pytest.approx(2 / 3)


def test_descriptor_product(orc):
    # src/query_disk_index.rs:135-142: integer sum of per-term truncations
    scales = np.array([1.0 / 512, -0.5 / 512], np.float32)
    desc = np.array([[0, 0], [255, 3]], np.uint8)
    want = int(np.float32(scales[0] * np.float32(255)) * np.float32(2 ** 32)) + int(np.float32(scales[1] * np.float32(3)) * np.float32(2 ** 32))
    assert orc.descriptor_product(scales, desc, 1) == want
    assert orc.descriptor_product(scales, desc, 0) == 0


def test_adc_is_sequential_fp32_sum(orc):
    g = np.load(os.path.join(GOLDEN, "pq_adc_4096.npz"))
    lut, codes = g["lut"], g["codes"]
    pq = orc.PQ(np.zeros((256, D), np.float32), np.eye(D, dtype=np.float32), 18, D)
    got = pq.asymmetric_dot_product(lut, codes)
    assert np.array_equal(got, g["adc"])
    # literal restatement of vector.rs:393-404 for a few vectors
    for j in (0, 1, 17, 4095):
        s = np.float32(0)
        for i in range(64):
            s = np.float32(s + lut[i, codes[j, i]])
        assert got[j] == int(np.float32(s * np.float32(2 ** 32)))
    assert np.array_equal(pq.adc_desc(lut, codes, g["desc"], g["scales"]), g["adc_desc"])


def test_quantize_first_max_wins_and_transform_orientation(orc):
    d, dpc = 64, 16
    # transform = permutation-like matrix so orientation T.x (not T^T.x) is visible (vector.rs:326)
    T = np.zeros((d, d), np.float32)
    for i in range(d):
        T[i, (i + 1) % d] = 1.0       # (T x)[i] = x[i+1]
    cents = np.zeros((4, d), np.float32)
    cents[1, :] = 1.0
    cents[2, :] = 1.0                 # duplicates of centroid 1 -> tie; first must win (:353-358)
    cents[3, :] = -1.0
    pq = orc.PQ(cents, T, dpc, d)
    x = np.arange(d, dtype=np.float32)[None, :]
    t = pq.apply_transform(x)
    assert np.array_equal(t[0], np.roll(x[0], -1))
    codes = pq.quantize_batch(np.ones((1, d), np.float32))
    assert list(codes[0]) == [1, 1, 1, 1]
    codes = pq.quantize_batch(-np.ones((1, d), np.float32))
    assert list(codes[0]) == [3, 3, 3, 3]
    codes = pq.quantize_batch(np.zeros((1, d), np.float32))   # all scores 0 -> first centroid
    assert list(codes[0]) == [0, 0, 0, 0]
    with pytest.raises(ValueError):
        orc.PQ(np.zeros((257, d), np.float32), T, dpc, d).quantize_batch(x)   # assert at :337


def test_preprocess_query_matches_definition(orc):
    rng = np.random.default_rng(2)
    d, dpc, nc = 128, 16, 8
    T = np.linalg.qr(rng.standard_normal((d, d)))[0].astype(np.float32)
    cents = rng.standard_normal((nc, d)).astype(np.float32)
    pq = orc.PQ(cents, T, dpc, d)
    q = rng.standard_normal(d).astype(np.float32)
    lut = pq.preprocess_query(q)
    t = (T.astype(np.float64) @ q.astype(np.float64))
    want = np.einsum("iu,jiu->ij", t.reshape(d // dpc, dpc), cents.astype(np.float64).reshape(nc, d // dpc, dpc))
    assert np.allclose(lut, want, rtol=1e-4, atol=1e-4)   # the transform itself is an fp32 GEMV


def test_select_shard_last_max(orc):
    # position_max_by_key returns the LAST maximum (src/query_disk_index.rs:254-256,447-450)
    cents = np.array([[1, 0], [0, 1], [1, 0]], np.float32)
    assert orc.select_shard(cents, np.array([1.0, 0.0], np.float32)) == 2
    assert orc.select_shard(cents, np.array([0.0, 1.0], np.float32)) == 1


def python_nb(cap):
    """Literal Python restatement of diskann/src/lib.rs:93-147 for cross-checking the C oracle."""
    state = {"ids": [], "scores": [], "visited": [], "next": None}

    def search(score):
        size = len(state["scores"])
        if size == 0:
            return 0
        base = 0
        while size > 1:
            half = size // 2
            mid = base + half
            if not (score > state["scores"][mid]):
                base = mid
            size -= half
        if score == state["scores"][base]:
            return base
        return base + (1 if score < state["scores"][base] else 0)

    def insert(idx, score):
        n = len(state["ids"])
        if n == cap and state["scores"][n - 1] > score:
            return
        loc = search(score)
        if loc < n and state["ids"][loc] == idx:
            return
        state["ids"].insert(loc, idx)
        state["scores"].insert(loc, score)
        state["visited"].insert(loc, 0)
        del state["ids"][cap:], state["scores"][cap:], state["visited"][cap:]
        state["next"] = loc if state["next"] is None else min(loc, state["next"])

    def next_unvisited():
        if state["next"] is None:
            return None
        cur = state["next"]
        old = cur
        state["visited"][cur] = 1
        while cur < len(state["ids"]) and state["visited"][cur]:
            cur += 1
        state["next"] = None if cur == len(state["ids"]) else cur
        return state["ids"][old]

    return state, insert, next_unvisited


def test_neighbour_buffer_trace(orc):
    g = np.load(os.path.join(GOLDEN, "neighbour_buffer_trace.npz"))
    nb = orc.NeighbourBuffer(int(g["cap"]))
    state, insert, next_unvisited = python_nb(int(g["cap"]))
    for step, (op, a, b) in enumerate(g["ops"]):
        if op == 0:
            nb.insert(int(a), int(b))
            insert(int(a), int(b))
        else:
            r = nb.next_unvisited()
            r2 = next_unvisited()
            assert (-1 if r is None else r) == int(a) == (-1 if r2 is None else r2)
        n = int(g["lens"][step])
        assert len(nb) == n == len(state["ids"])
        assert np.array_equal(nb.ids, g["ids"][step, :n]) and list(nb.ids) == state["ids"]
        assert np.array_equal(nb.scores, g["scores"][step, :n]) and list(nb.scores) == state["scores"]
        assert np.array_equal(nb.visited, g["visited"][step, :n])


def test_neighbour_buffer_rules(orc):
    nb = orc.NeighbourBuffer(3)
    for idx, s in ((1, 10), (2, 30), (3, 20)):
        nb.insert(idx, s)
    assert list(nb.ids) == [2, 3, 1]
    nb.insert(4, 5)            # full and worse than the last -> rejected (lib.rs:118)
    assert list(nb.ids) == [2, 3, 1]
    nb.insert(5, 10)           # full, EQUAL to the last -> proceeds, then truncated away or kept per search
    assert len(nb) == 3
    nb.insert(2, 30)           # same id found at its slot -> ignored (lib.rs:127-129)
    assert list(nb.ids)[:1] == [2]
    assert nb.next_unvisited() == 2 and nb.next_unvisited() == 3
    nb.insert(9, 100)          # better than everything: becomes next unvisited
    assert nb.next_unvisited() == 9


def test_greedy_search_visits_and_orders(orc):
    rng = np.random.default_rng(4)
    n, deg = 300, 8
    vecs = orc.gen_rows_f16(SEED_BASE, 0, n)
    adj = rng.integers(0, n, size=(n, deg), dtype=np.uint32)
    degs = np.full(n, deg, np.uint32)
    q = orc.gen_rows_f16(SEED_QUERY, 0, 1)[0]
    nb, dist = orc.greedy_search(vecs, adj, degs, 0, q, 32)
    s = nb.scores
    assert np.all(s[:-1] >= s[1:]) and dist > 0
    exact = orc.score_all(vecs, q)
    assert np.array_equal(exact[nb.ids], s)


def test_index_semantics(orc):
    rng = np.random.default_rng(6)
    x = (rng.standard_normal((500, 128)) / np.sqrt(128)).astype(np.float32)
    codes = orc.f16_bits(x)
    q = rng.standard_normal((3, 128)).astype(np.float32)
    d0, l0 = orc.index_search(codes, q, 7, order=0)
    d1, l1 = orc.index_search(codes, q, 7, order=1)
    assert np.array_equal(l0, l1)                      # both summation orders rank this data identically
    assert np.allclose(d0, d1, rtol=1e-5, atol=1e-6)
    want = orc.f16_to_f32(codes).astype(np.float64) @ q.astype(np.float64).T
    assert np.array_equal(np.argsort(-want[:, 0], kind="stable")[:7], l0[0])
    d, l = orc.index_search(codes[:3], q[:1], 5, order=0)
    assert list(l[0][3:]) == [-1, -1]


def test_total_embedding(orc):
    rng = np.random.default_rng(8)
    e = orc.f16_bits(rng.standard_normal((3, 64)).astype(np.float32))
    w = np.array([1.0, -0.5, 2.0], np.float32)
    got = orc.total_embedding(e, w)
    want = np.zeros(64, np.float32)
    for i in range(3):
        want += orc.f16_to_f32(e[i]) * w[i]
    assert np.array_equal(got, want)


def test_disk_greedy_search_against_literal_restatement(orc):
    """src/query_disk_index.rs:144-212 restated literally in Python (sets, lists) on a small index; pins the start
    score of 0, the per-iteration pre-buffer, both counters and the visited-list rule."""
    rng = np.random.default_rng(21)
    n, d, deg, L, nch = 300, 64, 6, 24, 4
    vecs = orc.f16_bits((rng.standard_normal((n, d)) / np.sqrt(d)).astype(np.float32))
    adj = rng.integers(0, n, size=(n, deg), dtype=np.uint32)
    degs = rng.integers(1, deg + 1, size=n).astype(np.uint32)
    codes = rng.integers(0, 16, size=(n, nch), dtype=np.uint8)
    lut = rng.standard_normal((nch, 16)).astype(np.float32)
    desc = rng.integers(0, 256, size=(n, 2), dtype=np.uint8)
    scales = np.array([0.001, -0.002], np.float32)
    has_url = (rng.random(n) > 0.2).astype(np.uint8)
    q = orc.f16_bits((rng.standard_normal(d) / np.sqrt(d)).astype(np.float32))

    def bias(i):
        return sum(int(orc.lib().orc_scale_dot_result(np.float32(scales[j]) * np.float32(desc[i, j]))) for j in range(2))

    def adc(i):
        s = np.float32(0)
        for c in range(nch):
            s = np.float32(s + lut[c, codes[i, c]])
        return int(orc.lib().orc_scale_dot_result(s))

    for beam, disable_pq in [(1, False), (3, False), (2, True)]:
        buf = orc.NeighbourBuffer(L)
        buf.insert(7, 0)
        visited_adjacent, visited, vlist, cmps, pq_cmps = {7}, set(), [], 0, 0
        while True:
            pts = []
            for _ in range(beam):
                p = buf.next_unvisited()
                if p is None:
                    break
                pts.append(p)
            if not pts:
                break
            pre = []
            for pt in pts:
                score = orc.fast_dot(q, vecs[pt]) + bias(pt)
                cmps += 1
                if pt not in visited:
                    visited.add(pt)
                    if has_url[pt]:
                        vlist.append((pt, score))
                for nb in adj[pt, :degs[pt]]:
                    if int(nb) not in visited_adjacent:
                        visited_adjacent.add(int(nb))
                        pre.append(int(nb))
                for nb in pre:
                    if disable_pq:
                        buf.insert(nb, orc.fast_dot(q, vecs[nb]) + bias(nb))
                    else:
                        buf.insert(nb, adc(nb) + bias(nb))
                        pq_cmps += 1
        got, vids, vsc, cm, pc = orc.disk_greedy_search(vecs, adj, degs, codes, desc, 7, q, lut, scales, disable_pq, beam, L,
                                                       has_url, n_centroids=16)
        assert (cm, pc) == (cmps, pq_cmps)
        assert np.array_equal(got.ids, buf.ids) and np.array_equal(got.scores, buf.scores)
        assert vids.tolist() == [v[0] for v in vlist] and vsc.tolist() == [v[1] for v in vlist]
        if beam > 1 and not disable_pq:
            assert pq_cmps > len(visited_adjacent) - 1          # the quirk: some neighbours are scored more than once


def test_medioid_running_mean(orc):
    rng = np.random.default_rng(22)
    vecs = orc.f16_bits((rng.standard_normal((37, 64)) / 8).astype(np.float32))
    c = np.zeros(64, np.float32)
    for i, v in enumerate(orc.f16_to_f32(vecs)):                   # lib.rs:55-58
        c = (c + (v - c) * np.float32(1.0 / np.float32(i + 1))).astype(np.float32)
    assert np.array_equal(orc.centroid_f16(vecs), orc.f16_bits(c))
    keys = [orc.dot_f64(v, orc.f16_bits(c)) for v in vecs]
    want = max(range(37), key=lambda i: (keys[i], i))             # last maximum
    assert orc.medioid(vecs) == want


def test_dedup_keep_rule(orc):
    # query_disk_index.rs:514-527: compared only against rows already KEPT, first occurrence wins
    d = 64
    e = np.eye(d, dtype=np.float32)
    a, b = e[0], e[1]
    near_a = (a * 0.96 + b * 0.28).astype(np.float32); near_a /= np.linalg.norm(near_a)
    near2 = (near_a * 0.96 + e[2] * 0.28).astype(np.float32); near2 /= np.linalg.norm(near2)     # close to near_a, not to a
    rows = orc.f16_bits(np.stack([a, near_a, near2, b, a]))
    sims = orc.f16_to_f32(rows) @ orc.f16_to_f32(rows).T
    assert sims[1, 0] > 0.95 and sims[2, 1] > 0.95 and sims[2, 0] < 0.95
    # near_a is dropped (dup of a); near2 is similar only to the DROPPED near_a, so it is kept; last row duplicates a
    assert orc.dedup_keep(rows).tolist() == [1, 0, 1, 1, 0]


def test_descriptor_bucket_is_rust_binary_search(orc):
    # dump_processor.rs:485-488 against a literal model of `binary_search_by`: equal element -> its index, else insertion point
    cdf = np.array([0.0, 0.1, 0.1, 0.1, 0.5, 0.9], np.float32)
    got = orc.descriptor_buckets(cdf[None, :], np.array([[-1.0], [0.0], [0.05], [0.3], [0.5], [0.95]], np.float32))[:, 0]
    assert got.tolist() == [0, 0, 1, 4, 4, 6]
    b = int(orc.descriptor_buckets(cdf[None, :], np.array([[0.1]], np.float32))[0, 0])
    assert cdf[b] == np.float32(0.1)                      # some index of the run of equal elements
    rng = np.random.default_rng(3)
    cdf = np.sort(rng.standard_normal(255).astype(np.float32))
    sc = rng.standard_normal((500, 1)).astype(np.float32)
    assert np.array_equal(orc.descriptor_buckets(cdf[None, :], sc)[:, 0], np.searchsorted(cdf, sc[:, 0], side="left").astype(np.uint8))
    assert int(orc.descriptor_buckets(cdf[None, :], np.array([[1e9]], np.float32))[0, 0]) == 255   # above the last quantile


def test_score_model_formula(orc):
    rng = np.random.default_rng(4)
    d, hdim, oc = 1152, 96, 3
    up = (rng.standard_normal((hdim, d)) / np.sqrt(d)).astype(np.float32)
    b = (0.1 * rng.standard_normal(hdim)).astype(np.float32)
    down = (rng.standard_normal((oc, hdim)) / np.sqrt(hdim)).astype(np.float32)
    x = rng.standard_normal((5, d)).astype(np.float32)
    h = (up.astype(np.float64) @ x.T.astype(np.float64)) + b[:, None]
    want = ((down.astype(np.float64) @ (h / (1 + np.exp(-h)))) * (d / hdim)).T        # score_model.rs:21-29
    assert np.allclose(orc.score_batch(up, b, down, x), want, rtol=2e-5, atol=2e-5)


# ---- Vamana build (diskann/src/lib.rs:183-389): literal Python restatement vs the C oracle ------------------------

I64_MIN = -(1 << 63)


def py_greedy_search(orc, vecs, lists, start, query, cap, base_only, qb):
    """lib.rs:183-211 with sets and lists; returns (buffer ids, visited_list)."""
    state, insert, next_unvisited = python_nb(cap)
    visited, visited_list = {start}, []
    insert(start, orc.fast_dot(query, vecs[start]))
    while True:
        pt = next_unvisited()
        if pt is None:
            break
        pre = []
        for nb in lists[pt]:
            fresh = nb not in visited
            visited.add(nb)
            if fresh and not (base_only and nb >= qb):
                pre.append(nb)
        for nb in pre:
            s = orc.fast_dot(query, vecs[nb])
            insert(nb, s)
            visited_list.append((nb, s))
    return state["ids"], visited_list


def py_robust_prune(orc, vecs, cands, p, cfg):
    """lib.rs:227-285, statement by statement (the sort is made stable: see the oracle's note on :233)."""
    cands = sorted(cands, key=lambda c: -c[1])[:cfg["maxc"]]
    cands = [list(c) for c in cands]
    neigh, ci = [], 0
    while len(neigh) < cfg["r"] and ci < len(cands):
        p_star, p_star_score = cands[ci]
        ci += 1
        if p_star == p or p_star_score == I64_MIN:
            continue
        neigh.append(p_star)
        scratch = [(i, cands[i][0]) for i in range(ci + 1, len(cands)) if cands[i][1] != I64_MIN]
        for i, p_prime in scratch:
            s = orc.fast_dot(vecs[p_prime], vecs[p_star])
            a = cfg["query_alpha"] if p_prime >= cfg["qb"] else cfg["alpha"]
            if (a * s) >> 16 >= cands[i][1]:
                cands[i][1] = I64_MIN
    if cfg["saturate"] or p >= cfg["qb"]:
        for cid, _ in cands:
            if len(neigh) == cfg["r"]:
                break
            if cid not in neigh:
                neigh.append(cid)
    return neigh


def py_build_graph(orc, vecs, lists, order, medioid, cfg):
    """lib.rs:287-324, the single-threaded form kept in its comments (:294,297)."""
    for sigma in order:
        is_query = sigma >= cfg["qb"]
        _, vl = py_greedy_search(orc, vecs, lists, medioid, vecs[sigma], cfg["l"], is_query, cfg["qb"])
        vl += [(n, orc.fast_dot(vecs[sigma], vecs[n])) for n in lists[sigma]]
        lists[sigma] = py_robust_prune(orc, vecs, vl, sigma, cfg)
        for nb in list(lists[sigma]):
            nn = lists[nb]
            if len(nn) == cfg["r"]:
                c = [(x, orc.fast_dot(vecs[nb], vecs[x])) for x in nn] + [(sigma, orc.fast_dot(vecs[nb], vecs[sigma]))]
                lists[nb] = py_robust_prune(orc, vecs, c, nb, cfg)
            elif sigma not in nn and len(nn) < cfg["r"]:
                nn.append(sigma)


def py_robust_stitch(orc, vecs, lists, queries_order, cfg):
    """lib.rs:326-374, queries taken one after another."""
    qb, n = cfg["qb"], len(lists)
    in_edges = [[] for _ in range(n - qb)]
    for b in range(qb):
        keep = []
        for e in lists[b]:
            if e >= qb:
                in_edges[e - qb].append(b)
            else:
                keep.append(e)
        lists[b] = keep
    for q in queries_order:
        for b in in_edges[q - qb]:
            cands = sorted([(x, orc.fast_dot(vecs[b], vecs[x])) for x in lists[q]], key=lambda c: -c[1])
            added = 0
            for x, _ in cands:
                if added >= cfg["max_add"] or len(lists[b]) >= cfg["r"]:
                    break
                if x in lists[b]:
                    continue
                lists[b].append(x)
                added += 1


def _as_lists(adj, deg):
    return [[int(x) for x in adj[i, :deg[i]]] for i in range(len(deg))]


@pytest.mark.parametrize("qb_frac,saturate,alpha", [(None, False, 65536), (0.85, False, 78643), (None, True, 65536)])
def test_build_graph_against_literal_restatement(orc, qb_frac, saturate, alpha):
    rng = np.random.default_rng(77)
    n, d, r = 260, 64, 6
    centres = rng.standard_normal((12, d))
    x = centres[rng.integers(0, 12, n)] + 0.6 * rng.standard_normal((n, d))
    vecs = orc.f16_bits((x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32))
    qb = int(n * qb_frac) if qb_frac else 0xFFFFFFFF
    adj, deg = orc.random_fill_graph(9, n, r)
    deg[3] = 2                 # a list that is not full: the push branch of the back-edge step
    adj[8, 1] = adj[8, 0]      # an id listed twice
    order = rng.permutation(n).astype(np.uint32)
    med = int(orc.medioid(vecs))
    cfg = dict(r=r, l=14, maxc=20, alpha=alpha, query_alpha=90000, saturate=saturate, qb=qb, max_add=2)
    oc = orc.BuildConfig.make(r=r, l=14, maxc=20, alpha=alpha, query_alpha=90000, saturate_graph=saturate, query_breakpoint=qb,
                              max_add_per_stitch_iter=2)
    lists = _as_lists(adj, deg)
    # one search + prune first, on the untouched graph
    nb, vi, vs = orc.greedy_search_visited(vecs, adj, deg, med, vecs[5], 14, False, qb)
    ids, vl = py_greedy_search(orc, vecs, lists, med, vecs[5], 14, False, qb)
    assert list(nb.ids) == ids and list(zip(vi.tolist(), vs.tolist())) == vl
    assert orc.robust_prune(vecs, vi, vs, 5, oc).tolist() == py_robust_prune(orc, vecs, vl, 5, cfg)
    # the whole pass
    py_build_graph(orc, vecs, lists, [int(v) for v in order], med, cfg)
    orc.build_graph(vecs, adj, deg, order, med, oc, 1)
    assert _as_lists(adj, deg) == lists
    assert deg.max() <= r
    if qb_frac:
        qorder = (qb + rng.permutation(n - qb)).astype(np.uint32)
        py_robust_stitch(orc, vecs, lists, [int(v) for v in qorder], cfg)
        orc.robust_stitch(vecs, adj, deg, qorder, oc)
        assert _as_lists(adj, deg) == lists


def test_robust_prune_known_answers(orc):
    """Hand-derivable cases: the candidate behind p_star escapes the test against it (lib.rs:240,250); p itself and
    discarded candidates are skipped; alpha relaxes the discard rule."""
    d = 64
    e = np.zeros((6, d), np.float32)
    e[0, 0] = 1.0                                   # p
    e[1, 0], e[1, 1] = 0.9, 0.1                     # closest to p
    e[2, 0], e[2, 1] = 0.8, 0.2                     # right behind it: never compared with candidate 1
    e[3, 0], e[3, 1] = 0.7, 0.3                     # dot with row 1 = 0.66 < 0.7: kept at alpha 1, discarded at alpha 1.2
    e[4, 0], e[4, 2] = 0.1, 0.9                     # dot with row 1 = 0.09 < 0.1: kept at alpha 1; 0.108 >= 0.1 at alpha 1.2
    e[5, 0], e[5, 1] = 0.6, 0.8                     # dot with row 1 = 0.62 >= 0.6: discarded
    vecs = orc.f16_bits(e)
    ids = np.array([1, 2, 3, 4, 5, 0], np.uint32)
    scores = orc.score_rows(vecs, ids, vecs[0])
    cfg = orc.BuildConfig.make(r=4, l=8, maxc=10)
    assert orc.robust_prune(vecs, ids, scores, 0, cfg).tolist() == [1, 2, 3, 4]
    cfg12 = orc.BuildConfig.make(r=4, l=8, maxc=10, alpha=78643)
    assert orc.robust_prune(vecs, ids, scores, 0, cfg12).tolist() == [1, 2]        # 2 survives only as the escapee
    sat = orc.BuildConfig.make(r=4, l=8, maxc=10, alpha=78643, saturate_graph=True)
    assert orc.robust_prune(vecs, ids, scores, 0, sat).tolist() == [1, 2, 0, 3]   # saturation refills in score order, p included
    small = orc.BuildConfig.make(r=4, l=8, maxc=2)
    assert orc.robust_prune(vecs, ids, scores, 0, small).tolist() == [1]           # maxc keeps (p, 1); p is skipped


def test_random_fill_graph_properties(orc):
    adj, deg = orc.random_fill_graph(4, 500, 12)
    assert (deg == 12).all() and adj.max() < 500 and all(len(set(r)) == 12 for r in adj)
    adj2, _ = orc.random_fill_graph(5, 500, 12)
    assert not np.array_equal(adj, adj2)
