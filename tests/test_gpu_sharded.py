"""The row-sharded search (BASELINE configs[3]) through its C-ABI entry points, against the ORACLE:
mse_shard_group (one process, a host thread per shard; logical shards share the one device of the test box),
mse_comm (RCCL all-gather; world of one here: the collective call path and the packed-block merge),
the threading contract of the ABI (include/mse.h: shared read-only base, one searcher per thread, thread-local errors)."""
import os
import threading

import numpy as np
import pytest

from conftest import SEED_BASE, SEED_QUERY

pytestmark = pytest.mark.gpu
D = 1152


@pytest.mark.parametrize("n,G,nq,k", [(40_003, 8, 24, 10), (20_000, 3, 140, 10), (5, 8, 3, 4), (4096, 2, 9, 70)])
def test_shard_group_logical_shards_match_oracle(gpu, mse, orc, n, G, nq, k):
    base = orc.gen_rows_f16(SEED_BASE, 0, n)
    q = orc.gen_rows_f16(SEED_QUERY, 0, nq)
    want_s, want_i = orc.bruteforce_topk(base, q, k)
    grp = mse.ShardGroup(G, D, devices=[0] * G)
    assert grp.n_shards == G and all(grp.device(g) == 0 and grp.peer_mapped(g) for g in range(G))
    grp.generate(SEED_BASE, 0, n)                      # every shard makes its own rows [lo, hi) on the device
    assert len(grp) == n
    for mode in (mse.MODE_MFMA, mse.MODE_EXACT):
        s, i = grp.bruteforce_topk(q, k, mode)
        assert np.array_equal(i, want_i) and np.array_equal(s, want_s), mode
    grp.load_host(base)                                # the same rows handed over as one host array
    s, i = grp.bruteforce_topk(q, k)
    assert np.array_equal(i, want_i) and np.array_equal(s, want_s)
    grp.close()


def test_shard_group_ties_across_shards_break_by_lower_global_id(gpu, mse, orc):
    # the same row in every shard: equal scores in different shards must come back in ascending global id
    row = orc.gen_rows_f16(SEED_BASE, 7, 1)
    base = np.repeat(row, 64, axis=0)
    q = orc.gen_rows_f16(SEED_QUERY, 0, 2)
    grp = mse.ShardGroup(4, D, devices=[0] * 4)
    grp.load_host(base)
    s, i = grp.bruteforce_topk(q, 10, mse.MODE_MFMA)
    assert i.tolist() == [list(range(10))] * 2
    ws, wi = orc.bruteforce_topk(base, q, 10)
    assert np.array_equal(s, ws) and np.array_equal(i, wi)
    grp.close()


def test_shard_group_without_peer_mappings(gpu, mse, orc, monkeypatch):
    """Devices that cannot map each other's memory take the staged path (queries copied to the shard's device, its block of
    records copied back by hipMemcpyPeerAsync): forced here on the one device, same answer."""
    monkeypatch.setenv("MSE_SHARD_NO_PEER", "1")
    n, G, nq, k = 30_001, 5, 150, 10
    base = orc.gen_rows_f16(SEED_BASE, 0, n)
    q = orc.gen_rows_f16(SEED_QUERY, 0, nq)
    want_s, want_i = orc.bruteforce_topk(base, q, k)
    grp = mse.ShardGroup(G, D, devices=[0] * G)
    assert not any(grp.peer_mapped(g) for g in range(G))
    grp.load_host(base)
    for mode in (mse.MODE_MFMA, mse.MODE_EXACT):
        s, i = grp.bruteforce_topk(q, k, mode)
        assert np.array_equal(i, want_i) and np.array_equal(s, want_s), mode
    grp.close()


def test_shard_group_rccl_exchange(gpu, mse, orc):
    """north_star's exchange inside ONE process: a communicator per shard device (ncclCommInitAll), one ncclAllGather of the packed
    records per search, each rank's collective issued by its shard's own host thread.  With one device visible the group of one
    shard is a world of one (the collective really runs); with several devices every device carries a shard.  Shards that share
    a device cannot form RCCL ranks: set_exchange fails cleanly and the group keeps answering over the peer-store exchange."""
    n_dev = gpu
    G = n_dev if n_dev > 1 else 1
    n, nq, k = 40_000 + 17, 130, 10
    rows = orc.gen_rows_f16(SEED_BASE, 0, n)
    q = orc.gen_rows_f16(SEED_QUERY, 0, nq)
    ws, wi = orc.bruteforce_topk(rows, q, k)
    grp = mse.ShardGroup(G, D, devices=list(range(G)))
    grp.load_host(rows)
    assert grp.exchange == grp.EXCHANGE_PEER and grp.rccl_ranks == 0
    s0, i0 = grp.bruteforce_topk(q, k, mse.MODE_MFMA)
    grp.set_exchange(grp.EXCHANGE_RCCL)
    assert grp.exchange == grp.EXCHANGE_RCCL and grp.rccl_ranks == G        # counted by RCCL, not by the caller
    for mode in (mse.MODE_MFMA, mse.MODE_EXACT):
        m = nq if mode == mse.MODE_MFMA else 8
        s1, i1 = grp.bruteforce_topk(q[:m], k, mode)
        assert np.array_equal(i1, wi[:m]) and np.array_equal(s1, ws[:m])
    assert np.array_equal(s0, ws) and np.array_equal(i0, wi)
    t = grp.last_timing()
    assert t["wall_ms"] > 0 and t["local_search_ms"] > 0 and t["exchange_ms"] >= 0 and t["merge_ms"] > 0
    assert t["local_search_ms"] + t["merge_ms"] <= t["wall_ms"] * 1.05
    grp.set_exchange(grp.EXCHANGE_PEER)                                      # and back
    s2, i2 = grp.bruteforce_topk(q, k, mse.MODE_MFMA)
    assert np.array_equal(s2, ws) and np.array_equal(i2, wi)
    grp.close()
    # logical shards (two on one device) cannot be RCCL ranks: a labelled failure, never a crash, and the group still works
    two = mse.ShardGroup(2, D, devices=[0, 0])
    two.load_host(rows)
    with pytest.raises(mse.MseError, match="own device"):
        two.set_exchange(two.EXCHANGE_RCCL)
    assert two.exchange == two.EXCHANGE_PEER and two.rccl_ranks == 0
    s3, i3 = two.bruteforce_topk(q, k, mse.MODE_MFMA)
    assert np.array_equal(s3, ws) and np.array_equal(i3, wi)
    two.close()


def test_bench_rccl_probe_runs_in_a_child_process(gpu):
    """bench.py --gpus N probes the in-process RCCL exchange in a child process under a timeout before relying on it; here the
    same probe on as many devices as are visible (one device: a world of one)."""
    import subprocess
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    pr = subprocess.run([sys.executable, "-c", bench.rccl_probe_code(gpu)], capture_output=True, text=True, timeout=300)
    assert "PROBE_OK" in pr.stdout, pr.stderr[-500:]


def test_shard_group_errors(gpu, mse):
    with pytest.raises(mse.MseError):
        mse.ShardGroup(2, D, devices=[0, 99])
    with pytest.raises(mse.MseError):
        mse.ShardGroup(2, 100)
    grp = mse.ShardGroup(2, D, devices=[0, 0])
    with pytest.raises(mse.MseError, match="holds no rows"):
        grp.bruteforce_topk(np.zeros((1, D), np.uint16), 3)
    grp.close()


def test_rccl_comm_world_of_one(gpu, mse, orc):
    """mse_comm on the one GPU of the test box: ncclGetUniqueId / ncclCommInitRank / ncclAllGather really run (RCCL reports
    its own rank count), and the packed-block merge returns the oracle's answer.  More ranks need more GPUs (bench.py --gpus N)."""
    import torch
    n, nq, k = 30_000, 20, 10
    base = orc.gen_rows_f16(SEED_BASE, 1000, n)
    q = orc.gen_rows_f16(SEED_QUERY, 0, nq)
    ws, wi = orc.bruteforce_topk(base, q, k)
    comm = mse.Comm(mse.Comm.unique_id(), 0, 1)
    assert comm.size == 1 and comm.rank == 0
    sr = mse.Searcher(mse.VectorList.generate(SEED_BASE, 1000, n))
    qd = torch.from_numpy(q.view(np.int16)).cuda()
    out_s = torch.empty((nq, k), dtype=torch.int64, device="cuda")
    out_i = torch.empty((nq, k), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    comm.search_dev(sr, qd.data_ptr(), nq, k, out_s.data_ptr(), out_i.data_ptr(), mse.MODE_MFMA, id_offset=5000)
    torch.cuda.synchronize()
    assert np.array_equal(out_s.cpu().numpy(), ws)
    assert np.array_equal(out_i.cpu().numpy().view(np.uint32), wi + np.uint32(5000))
    comm.close()


def test_packed_merge_of_eight_blocks_matches_host_merge(gpu, mse, orc):
    """What eight ranks' all-gather would deliver: eight packed blocks, merged on the device, against the numpy merge."""
    import ctypes as C
    import torch
    from mse import ffi, shard
    G, nq, k = 8, 33, 10
    rng = np.random.default_rng(5)
    B = int(ffi.lib().mse_topk_block_bytes(nq, k))
    assert B % 16 == 0 and B >= nq * k * 12
    blocks = np.zeros((G, B), np.uint8)
    sc = rng.integers(-2 ** 40, 2 ** 40, size=(G, nq, k)).astype(np.int64)
    sc[:, :, ::3] = 12345                                  # plenty of equal scores across shards
    ids = rng.permutation(G * nq * k).astype(np.uint32).reshape(G, nq, k)
    ids[3, :, 5:] = 0xFFFFFFFF                             # a shard with fewer than k rows
    for g in range(G):
        blocks[g, :nq * k * 8] = sc[g].view(np.uint8).reshape(-1)
        blocks[g, nq * k * 8:nq * k * 12] = ids[g].view(np.uint8).reshape(-1)
    want_s, want_i = shard.merge_topk_numpy(sc.transpose(1, 0, 2).reshape(nq, G * k), ids.transpose(1, 0, 2).reshape(nq, G * k), k)
    sr = mse.Searcher(mse.VectorList.generate(SEED_BASE, 0, 64))
    bd = torch.from_numpy(blocks).cuda()
    out_s = torch.empty((nq, k), dtype=torch.int64, device="cuda")
    out_i = torch.empty((nq, k), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    ffi.check(ffi.lib().mse_merge_topk_packed_dev(sr._h, bd.data_ptr(), G, nq, k, out_s.data_ptr(), out_i.data_ptr()))
    ffi.check(ffi.lib().mse_device_synchronize())
    assert np.array_equal(out_s.cpu().numpy(), want_s)
    assert np.array_equal(out_i.cpu().numpy().view(np.uint32), want_i)


def test_concurrent_searchers_share_one_base(gpu, mse, orc):
    """include/mse.h threading contract (the reference: a thread per core, each with its own Scratch over shared Arc maps,
    src/query_disk_index.rs:714-731): two threads, two searchers, ONE base, different queries, at the same time."""
    n, nq, k = 60_000, 40, 10
    vl = mse.VectorList.generate(SEED_BASE, 0, n)
    base = orc.gen_rows_f16(SEED_BASE, 0, n)
    qs = [orc.gen_rows_f16(SEED_QUERY, 100 * t, nq) for t in range(2)]
    want = [orc.bruteforce_topk(base, q, k) for q in qs]
    got, errs = [None, None], []
    start = threading.Barrier(2)

    def work(t):
        try:
            s = mse.Searcher(vl)
            start.wait()
            for rep in range(6):
                mode = mse.MODE_MFMA if (rep + t) % 2 else mse.MODE_EXACT
                r = s.bruteforce_topk(qs[t][:8] if mode == mse.MODE_EXACT else qs[t], k, mode)
                m = 8 if mode == mse.MODE_EXACT else nq
                if not (np.array_equal(r[0], want[t][0][:m]) and np.array_equal(r[1], want[t][1][:m])):
                    errs.append((t, rep, mode))
            got[t] = s.bruteforce_topk(qs[t], k, mse.MODE_MFMA)
            s.close()
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    th = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    [x.start() for x in th]
    [x.join() for x in th]
    assert not errs, errs
    for t in range(2):
        assert np.array_equal(got[t][0], want[t][0]) and np.array_equal(got[t][1], want[t][1])


def test_last_error_is_thread_local(gpu, mse):
    from mse import ffi
    L = ffi.lib()
    seen = {}

    def bad():
        assert L.mse_base_wrap_device(None, 10, 100) is None      # width not a multiple of 64 -> error on THIS thread
        seen["bad"] = ffi.last_error()

    t = threading.Thread(target=bad)
    assert L.mse_scale_dot_f32(1.0) == 2 ** 32
    before = ffi.last_error()
    t.start()
    t.join()
    assert "multiple of 64" in seen["bad"]
    assert ffi.last_error() == before                                # the failing thread's message did not leak here


@pytest.mark.parametrize("staged", [False, True])
def test_config4_full_size_1e8_eight_shards(gpu, mse, orc, monkeypatch, staged):
    """BASELINE configs[3]: the 1e8 x 1152 index sharded 8 ways (12.5 M rows = 28.8 GB per shard), here as eight logical shards
    on the one device (230 GB resident), through mse_shard_group.  The oracle cannot scan 1e8 rows in a test, so:
    size-independent properties, with every returned score re-derived by the oracle from regenerated rows.
    staged: MSE_SHARD_NO_PEER=1 -- every shard takes the path of a device that cannot map the root's memory (queries copied in,
    packed block copied back), at full size."""
    from mse import ffi
    if staged:
        monkeypatch.setenv("MSE_SHARD_NO_PEER", "1")
    free_b, total_b = ffi.sz(), ffi.sz()
    ffi.check(ffi.lib().mse_device_mem_info(free_b, total_b))
    n, G, nq, k = 100_000_000, 8, 136, 10
    if free_b.value < n * D * 2 + (24 << 30):
        pytest.skip(f"needs {n * D * 2 / 1e9:.0f} GB of free HBM, {free_b.value / 1e9:.0f} GB free")
    grp = mse.ShardGroup(G, D, devices=[0] * G)
    assert sum(grp.peer_mapped(g) for g in range(G)) == (0 if staged else G)
    grp.generate(SEED_BASE, 0, n)
    assert len(grp) == n
    q = orc.gen_rows_f16(SEED_QUERY, 0, nq)
    # planted winners: copies of rows at shard seams and deep inside shards (global ids carry the shard offset)
    planted = [0, 12_499_999, 12_500_000, 99_999_999, 87_654_321, 50_000_000]
    for j, r in enumerate(planted):
        q[j] = orc.gen_rows_f16(SEED_BASE, r, 1)[0]
    sm, im = grp.bruteforce_topk(q, k, mse.MODE_MFMA)
    se, ie = grp.bruteforce_topk(q[:8], k, mse.MODE_EXACT)
    assert np.array_equal(im[:8], ie) and np.array_equal(sm[:8], se)            # two independent kernels agree, merged
    assert [int(im[j, 0]) for j in range(len(planted))] == planted              # a row is its own best match
    assert np.all(sm[:, :-1] >= sm[:, 1:])                                      # sorted
    assert np.all(im < n) and all(len(set(r)) == k for r in im.tolist())        # valid, distinct global ids
    assert len({int(x) // 12_500_000 for x in im.reshape(-1)}) == G             # winners come from every shard
    for qi in (0, 5, 77, 135):                                                  # returned scores are the oracle's
        for j in (0, 3, 9):
            row = orc.gen_rows_f16(SEED_BASE, int(im[qi, j]), 1)[0]
            assert orc.fast_dot(q[qi], row) == int(sm[qi, j])
    rng = np.random.default_rng(3)                                              # the k-th score bounds sampled outsiders
    sample = rng.integers(0, n, 400)
    rows = np.stack([orc.gen_rows_f16(SEED_BASE, int(r), 1)[0] for r in sample])
    for qi in (5, 77):
        sc = orc.score_all(rows, q[qi])
        assert np.all(sc[~np.isin(sample, im[qi])] <= sm[qi, -1])
    sm2, im2 = grp.bruteforce_topk(q, k, mse.MODE_MFMA)                          # idempotence
    assert np.array_equal(sm, sm2) and np.array_equal(im, im2)
    assert all(grp.searcher(g).last_stats()["widened_queries"] == 0 for g in range(G))
    grp.close()


def test_borrowed_base_rows_changed(gpu, mse, orc):
    """A wrapped (borrowed) base caches its largest row norm -- the bound behind the MFMA scan's certificate.  After the caller
    rewrites rows (here: scaled up 40x, far outside the cached bound) mse_base_rows_changed makes the next search measure it
    again, and the batched answer equals the oracle's on the new contents."""
    import torch
    from mse import ffi
    n, nq, k = 20_000, 24, 10
    rows = orc.gen_rows_f16(SEED_BASE, 0, n)
    t = torch.from_numpy(rows.view(np.int16).copy()).cuda()
    vl = mse.VectorList.wrap_device(t.data_ptr(), n, D, keepalive=t)
    s = mse.Searcher(vl)
    q = orc.gen_rows_f16(SEED_QUERY, 0, nq)
    ws, wi = orc.bruteforce_topk(rows, q, k)
    got = s.bruteforce_topk(q, k, mse.MODE_MFMA)
    assert np.array_equal(got[0], ws) and np.array_equal(got[1], wi)
    big = (orc.f16_to_f32(rows) * np.float32(40.0)).astype(np.float16).view(np.uint16)
    big[::7] = rows[::7]                                   # mixed magnitudes: near ties between small and large rows matter
    t.copy_(torch.from_numpy(big.view(np.int16)).cuda())
    torch.cuda.synchronize()
    ffi.check(ffi.lib().mse_base_rows_changed(vl._h))
    ws2, wi2 = orc.bruteforce_topk(big, q, k)
    got2 = s.bruteforce_topk(q, k, mse.MODE_MFMA)
    assert np.array_equal(got2[0], ws2) and np.array_equal(got2[1], wi2)
    assert not np.array_equal(wi, wi2)


# ---- the approximate-search paths over the same shards (round 5) --------------------------------------------------------------

def _ann_fixture(mse, orc, n, seed):
    from test_gpu_pq_index_graph import clustered_rows, train_pq
    x = clustered_rows(orc, n, n_centres=40, seed=seed)
    base = orc.f16_bits(x)
    cents, T = train_pq(orc, x[:2000], iters=2)
    return x, base, cents, T


@pytest.mark.parametrize("no_peer", [False, True])
def test_sharded_pq_scan_equals_the_unsharded_call_bit_for_bit(gpu, mse, orc, monkeypatch, no_peer):
    """configs[4] over 8 logical shards (codes + descriptors + rows partitioned by shard_range): ADC top-r of every shard -> exchange
    -> the index's top-r -> every shard re-scores ITS members exactly -> exchange -> top-k.  The merged result equals the unsharded
    mse_pq_scan_topk_batch call bit for bit -- ids and i64 scores -- with and without descriptor scales, over both exchange paths
    (peer stores; staged copies as for devices without a peer mapping), for batches that use the eight-, four- and two-query scans."""
    if no_peer:
        monkeypatch.setenv("MSE_SHARD_NO_PEER", "1")
    rng = np.random.default_rng(51)
    n, G, r, k = 20_011, 8, 60, 10
    x, base, cents, T = _ann_fixture(mse, orc, n, 11)
    desc = rng.integers(0, 256, size=(n, 4), dtype=np.uint8)
    pq = mse.ProductQuantizer(cents, T, 18, D)
    vecs = mse.VectorList.from_f16s(base, D)
    whole_codes = mse.Codes.quantize_base(pq, vecs, desc)
    whole = mse.Searcher(vecs)
    grp = mse.ShardGroup(G, D, devices=[0] * G)
    grp.load_host(base)
    keep = []
    for g in range(G):
        lo, hi = mse.shard_range(n, g, G)
        assert grp.first_row(g) == lo and len(grp.base(g)) == hi - lo
        codes_g = mse.Codes.quantize_base(pq, grp.base(g), desc[lo:hi])       # the shard's own codes, made from ITS resident rows
        grp.attach_pq(g, pq, codes_g)
        keep.append(codes_g)
    from test_gpu_pq_index_graph import clustered_rows
    qs = (clustered_rows(orc, 23, n_centres=40, seed=500) * np.float32(1.2)).astype(np.float32)
    scales = np.array([0.5, 0, -0.25, 1.0], np.float32) / np.float32(512)
    for sc in (None, scales):
        for nq in (23, 3, 1):
            want_s, want_i = pq.scan_topk_batch(whole_codes, qs[:nq], r, k, whole, sc)
            got_s, got_i = grp.pq_scan_topk(qs[:nq], r, k, sc)
            assert np.array_equal(got_i, want_i) and np.array_equal(got_s, want_s), (sc is not None, nq)
    # and against the oracle's statement of configs[4] for one query: ADC over all codes, top-r (lower id on ties), exact re-score, top-k
    opq = orc.PQ(cents, T, 18, D)
    codes_h = opq.quantize_batch(orc.f16_to_f32(base))
    adc = opq.asymmetric_dot_product(opq.preprocess_query(qs[0]), codes_h)
    top_r = np.lexsort((np.arange(n), -adc))[:r]
    exact = np.array([orc.fast_dot(orc.f16_bits(qs[0]), base[i]) for i in top_r], np.int64)
    order = np.lexsort((top_r, -exact))[:k]
    got_s, got_i = grp.pq_scan_topk(qs[:1], r, k, None)
    assert np.array_equal(got_i[0], top_r[order].astype(np.uint32)) and np.array_equal(got_s[0], exact[order])
    with pytest.raises(mse.MseError):
        g2 = mse.ShardGroup(2, D, devices=[0, 0])
        g2.load_host(base[:100])
        g2.pq_scan_topk(qs[:2], r, k)                              # nothing attached
    grp.close()


def test_sharded_graph_query_equals_the_merge_of_per_shard_oracle_searches(gpu, mse, orc):
    """One graph per shard over its own rows (the reference's shards), 8 logical shards: every shard answers the batch from ITS graph
    with ITS entry table, one exchange, merge by (score desc, id asc).  Against the oracle: per shard the oracle's entry choice +
    disk_greedy_search + sort of the visited list, ids lifted by the shard's first row, merged on the host."""
    from test_gpu_pq_index_graph import clustered_rows, knn_graph
    rng = np.random.default_rng(52)
    n, G, deg, nq, k, L = 12_000, 8, 12, 21, 10, 24
    x = clustered_rows(orc, n, n_centres=40, seed=12)
    base = orc.f16_bits(x)
    qh = orc.f16_bits(clustered_rows(orc, nq, n_centres=40, seed=501))
    grp = mse.ShardGroup(G, D, devices=[0] * G)
    grp.load_host(base)
    shards = []
    for g in range(G):
        lo, hi = mse.shard_range(n, g, G)
        adj, degs = knn_graph(x[lo:hi], deg, rng)
        dg = mse.DeviceGraph(mse.IndexGraph(adj, degs))
        entries = np.sort(rng.choice(hi - lo, 16, replace=False)).astype(np.uint32)
        mse.set_entries(dg, grp.base(g), entries)
        grp.attach_graph(g, dg)
        shards.append((lo, hi, adj, degs, entries, dg))
    got_s, got_i = grp.query_topk(qh, k, None, None, True, 2, L)
    all_s = np.full((nq, G * k), np.iinfo(np.int64).min, np.int64)
    all_i = np.full((nq, G * k), 0xFFFFFFFF, np.uint32)
    for g, (lo, hi, adj, degs, entries, _) in enumerate(shards):
        rows = base[lo:hi]
        _, best = orc.bruteforce_topk(rows[entries], qh, 1)
        for q in range(nq):
            _, vids, vsc, _, _ = orc.disk_greedy_search(rows, adj, degs, np.zeros((hi - lo, 64), np.uint8), np.zeros((hi - lo, 4), np.uint8),
                                                         int(entries[best[q, 0]]), qh[q], np.zeros(64 * 256, np.float32), None, True, 2, L, None)
            order = sorted(range(len(vids)), key=lambda j: (-int(vsc[j]), int(vids[j])))[:k]
            all_s[q, g * k:g * k + len(order)] = vsc[order]
            all_i[q, g * k:g * k + len(order)] = vids[order] + np.uint32(lo)
    want_s, want_i = mse.shard.merge_topk_numpy(all_s, all_i, k)
    assert np.array_equal(got_i, want_i) and np.array_equal(got_s, want_s)
    grp.close()


def test_comm_world_of_one_ann_paths(gpu, mse, orc):
    """The one-process-per-GPU forms (mse_comm_pq_scan_topk / mse_comm_query_topk / mse_comm_exchange_dev) as a world of one: the
    collective call path, both exchanges of the PQ protocol and the packed-block merge run, and the answers are the single-GPU calls'."""
    import torch
    from test_gpu_pq_index_graph import clustered_rows, knn_graph
    rng = np.random.default_rng(53)
    n, r, k, nq = 6000, 50, 10, 12
    x, base, cents, T = _ann_fixture(mse, orc, n, 13)
    pq = mse.ProductQuantizer(cents, T, 18, D)
    vecs = mse.VectorList.from_f16s(base, D)
    s = mse.Searcher(vecs)
    codes = mse.Codes.quantize_base(pq, vecs)
    qs = clustered_rows(orc, nq, n_centres=40, seed=502).astype(np.float32)
    comm = mse.Comm(mse.Comm.unique_id(), 0, 1)
    out_s = torch.empty((nq, k), dtype=torch.int64, device="cuda")
    out_i = torch.empty((nq, k), dtype=torch.int32, device="cuda")
    comm.pq_scan_topk(pq, codes, s, qs, r, k, 1000, out_s.data_ptr(), out_i.data_ptr())        # rank's first row 1000: ids come back lifted
    want_s, want_i = pq.scan_topk_batch(codes, qs, r, k, s)
    assert np.array_equal(out_s.cpu().numpy(), want_s) and np.array_equal(out_i.cpu().numpy().view(np.uint32), want_i + np.uint32(1000))
    adj, degs = knn_graph(x, 12, rng)
    dg = mse.DeviceGraph(mse.IndexGraph(adj, degs))
    mse.set_entries(dg, vecs, np.sort(rng.choice(n, 20, replace=False)).astype(np.uint32))
    qh = orc.f16_bits(qs)
    comm.query_topk(s, dg, qh, k, 77, out_s.data_ptr(), out_i.data_ptr(), disable_pq=True, beamwidth=2, search_list=24)
    wi, ws, _ = mse.disk_query_topk(s, None, None, dg, qh, k, None, None, None, True, 2, 24)
    assert np.array_equal(out_s.cpu().numpy(), ws) and np.array_equal(out_i.cpu().numpy().view(np.uint32), np.where(wi == 0xFFFFFFFF, wi, wi + np.uint32(77)))
    comm.close()
