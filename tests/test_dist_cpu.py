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
