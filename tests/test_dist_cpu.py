"""world_size-2 gloo test of the sharded search plumbing (mse/shard.py): row partition, global id
offsets, ONE all-gather of per-shard top-k records, k-way merge.  No GPU here, so each rank's
local top-k comes from the CPU oracle (test infrastructure); on the GPU box the same plumbing is
fed by the HIP path (tests/test_gpu_bruteforce.py::test_sharded_equals_whole)."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, SEED_BASE, SEED_QUERY


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_rows, nq, k, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    for p in (ROOT, os.path.join(ROOT, "meme-search-engine_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    from oracle import orc
    from mse import shard
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard.shard_range(n_rows, rank, world)
    base = orc.gen_rows_f16(SEED_BASE, lo, hi - lo, 128)
    queries = orc.gen_rows_f16(SEED_QUERY, 0, nq, 128)
    s, i = orc.bruteforce_topk(base, queries, k)
    gid = np.where(i == 0xFFFFFFFF, i, i + np.uint32(lo)).astype(np.uint32)
    ls = torch.from_numpy(s)
    li = torch.from_numpy(gid.astype(np.int64))
    ms, mi = shard.all_gather_topk(ls, li, k)
    dist.barrier()
    if rank == 0:
        np.savez(os.path.join(out_dir, "merged.npz"), s=ms.numpy(), i=mi.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("n_rows,k", [(1000, 10), (5, 4)])
def test_two_rank_sharded_topk_equals_whole(orc, tmp_path, n_rows, k):
    import torch.multiprocessing as mp
    nq = 3
    port = _free_port()
    mp.spawn(_worker, args=(2, port, n_rows, nq, k, str(tmp_path)), nprocs=2, join=True)
    got = np.load(tmp_path / "merged.npz")
    base = orc.gen_rows_f16(SEED_BASE, 0, n_rows, 128)
    queries = orc.gen_rows_f16(SEED_QUERY, 0, nq, 128)
    s, i = orc.bruteforce_topk(base, queries, k)
    assert np.array_equal(got["s"], s)
    assert np.array_equal(got["i"].astype(np.uint32), i)


def _ann_worker(rank, world, port, n_rows, nq, r, k, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    for p in (ROOT, os.path.join(ROOT, "meme-search-engine_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    from oracle import orc
    from mse import shard
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard.shard_range(n_rows, rank, world)
    base, codes, desc, queries, opq, scales = _ann_data(orc, n_rows, nq)
    luts = [opq.preprocess_query(orc.f16_to_f32(queries[q])) for q in range(nq)]

    def local_adc_topr():
        s = np.full((nq, r), np.iinfo(np.int64).min, np.int64)
        i = np.full((nq, r), 0xFFFFFFFF, np.uint32)
        for q in range(nq):
            adc = opq.adc_desc(luts[q], codes[lo:hi], desc[lo:hi], scales)
            order = np.lexsort((np.arange(hi - lo), -adc))[:r]
            s[q, :order.size] = adc[order]
            i[q, :order.size] = order
        return s, i

    def local_exact(q, local_id):
        return int(orc.fast_dot(queries[q], base[lo + local_id])) + int(orc.descriptor_product(scales, desc[lo:hi], local_id))

    ms, mi = shard.two_phase_pq_scan(local_adc_topr, local_exact, lo, hi, r, k)
    # the graph path's exchange: every rank's own [nq, k] records with global ids, one all-gather, merge
    ls = np.stack([np.sort(orc.bruteforce_topk(base[lo:hi], queries[q:q + 1], k)[0][0])[::-1] for q in range(nq)])   # any per-rank records will do
    li = np.stack([orc.bruteforce_topk(base[lo:hi], queries[q:q + 1], k)[1][0].astype(np.int64) + lo for q in range(nq)])
    gs, gi = shard.all_gather_topk(torch.from_numpy(np.ascontiguousarray(ls)), torch.from_numpy(li), k)
    dist.barrier()
    if rank == 0:
        np.savez(os.path.join(out_dir, "ann.npz"), s=ms.numpy(), i=mi.numpy(), gs=gs.numpy(), gi=gi.numpy())
    dist.destroy_process_group()


def _ann_data(orc, n_rows, nq):
    rng = np.random.default_rng(5)
    d = 128
    base = orc.gen_rows_f16(SEED_BASE, 0, n_rows, d)
    queries = orc.gen_rows_f16(SEED_QUERY, 0, nq, d)
    cents = (rng.standard_normal((256, d)) / np.sqrt(d)).astype(np.float32)
    T = np.linalg.qr(rng.standard_normal((d, d)))[0].astype(np.float32)
    opq = orc.PQ(cents, T, 2, d)                     # 64 chunks of 2 dims
    codes = opq.quantize_batch(orc.f16_to_f32(base))
    desc = rng.integers(0, 256, size=(n_rows, 4), dtype=np.uint8)
    scales = (np.array([0.5, 0, -0.25, 1.0], np.float32) / np.float32(512))
    return base, codes, desc, queries, opq, scales


@pytest.mark.parametrize("n_rows,r,k", [(700, 40, 10), (9, 6, 4)])
def test_two_rank_sharded_pq_scan_and_graph_exchange(orc, tmp_path, n_rows, r, k):
    """The protocol of the sharded approximate paths on gloo, world 2: the two-phase PQ scan (ADC top-r of every rank -> all-gather ->
    the index's top-r -> exact re-score of each rank's members -> all-gather -> top-k) equals the oracle's unsharded scan + re-rank;
    per-rank [nq, k] records with global ids merge to the whole index's top-k (the graph path's one exchange)."""
    import torch.multiprocessing as mp
    nq = 3
    port = _free_port()
    mp.spawn(_ann_worker, args=(2, port, n_rows, nq, r, k, str(tmp_path)), nprocs=2, join=True)
    got = np.load(tmp_path / "ann.npz")
    base, codes, desc, queries, opq, scales = _ann_data(orc, n_rows, nq)
    for q in range(nq):
        adc = opq.adc_desc(opq.preprocess_query(orc.f16_to_f32(queries[q])), codes, desc, scales)
        top_r = np.lexsort((np.arange(n_rows), -adc))[:r]
        exact = np.array([int(orc.fast_dot(queries[q], base[i])) + int(orc.descriptor_product(scales, desc, int(i))) for i in top_r], np.int64)
        order = np.lexsort((top_r, -exact))[:k]
        m = order.size
        assert np.array_equal(got["i"][q, :m], top_r[order]) and np.array_equal(got["s"][q, :m], exact[order]), q
        assert np.all(got["i"][q, m:] == 0xFFFFFFFF)
    s, i = orc.bruteforce_topk(base, queries, k)
    assert np.array_equal(got["gs"], s) and np.array_equal(got["gi"].astype(np.uint32), i)
