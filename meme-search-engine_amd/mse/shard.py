"""Row-sharded brute-force search across the GPUs of one node.

The reference has no multi-GPU code; scoring is row-independent, so the base
rows are partitioned contiguously, every rank scores the same query batch against its shard and
returns ids offset to global ids, and ONE all-gather of the per-shard top-k records
([nq, k] i64 scores + u32 ids, a few KiB) over RCCL/xGMI followed by a k-way merge gives the
global result.  No other collective exists on this path.

torch.distributed is plumbing here (process group + all_gather); the merge of the gathered
records runs in the HIP selection kernel when the tensors are on the device.
"""
import numpy as np


def shard_range(n_rows, rank, world):
    """Contiguous split, remainder spread over the first ranks: rows [lo, hi) for `rank`."""
    base, rem = divmod(n_rows, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def merge_topk_numpy(scores, ids, k):
    """scores/ids: [nq, m] candidate records -> [nq, k] best by (score desc, id asc).
    Records with id 0xFFFFFFFF are empty."""
    scores = np.asarray(scores, np.int64)
    ids = np.asarray(ids, np.uint32)
    nq = scores.shape[0]
    out_s = np.full((nq, k), np.iinfo(np.int64).min, np.int64)
    out_i = np.full((nq, k), 0xFFFFFFFF, np.uint32)
    for q in range(nq):
        valid = ids[q] != 0xFFFFFFFF
        s, i = scores[q][valid], ids[q][valid]
        order = np.lexsort((i, -s))[:k]
        out_s[q, :order.size] = s[order]
        out_i[q, :order.size] = i[order]
    return out_s, out_i


def merge_topk_torch(scores, ids, k):
    """Same contract on torch tensors (any device): scores int64 [nq, m], ids int64/int32 [nq, m]
    holding u32 values; empty records carry id 0xFFFFFFFF."""
    import torch
    ids64 = ids.to(torch.int64) & 0xFFFFFFFF
    empty = ids64 == 0xFFFFFFFF
    s = torch.where(empty, torch.full_like(scores, torch.iinfo(torch.int64).min), scores)
    # stable two-pass sort: by id ascending, then by score descending (stable keeps id order in ties)
    o1 = torch.argsort(ids64, dim=1, stable=True)
    s1 = torch.gather(s, 1, o1)
    i1 = torch.gather(ids64, 1, o1)
    e1 = torch.gather(empty, 1, o1)
    o2 = torch.argsort(s1, dim=1, descending=True, stable=True)
    s2 = torch.gather(s1, 1, o2)[:, :k]
    i2 = torch.gather(i1, 1, o2)[:, :k]
    e2 = torch.gather(e1, 1, o2)[:, :k]
    i2 = torch.where(e2, torch.full_like(i2, 0xFFFFFFFF), i2)
    if s2.shape[1] < k:
        pad = k - s2.shape[1]
        s2 = torch.cat([s2, torch.full((s2.shape[0], pad), torch.iinfo(torch.int64).min, dtype=s2.dtype, device=s2.device)], 1)
        i2 = torch.cat([i2, torch.full((i2.shape[0], pad), 0xFFFFFFFF, dtype=i2.dtype, device=i2.device)], 1)
    return s2, i2


def all_gather_topk(local_scores, local_ids, k, group=None):
    """local_*: torch tensors [nq, k] (scores int64, ids int32 viewed as u32, already global ids).
    One all_gather each, then the k-way merge.  Returns (scores [nq,k] int64, ids [nq,k] int64)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    gs = [torch.empty_like(local_scores) for _ in range(world)]
    gi = [torch.empty_like(local_ids) for _ in range(world)]
    dist.all_gather(gs, local_scores.contiguous(), group=group)
    dist.all_gather(gi, local_ids.contiguous(), group=group)
    return merge_topk_torch(torch.cat(gs, 1), torch.cat(gi, 1), k)
