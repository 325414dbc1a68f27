"""Row-sharded brute-force search across the GPUs of one node.

The reference has no multi-GPU code; scoring is row-independent, so the base
rows are partitioned contiguously, every rank scores the same query batch against its shard and
returns ids offset to global ids, and ONE all-gather of the per-shard top-k records
([nq, k] i64 scores + u32 ids, a few KiB) over RCCL/xGMI followed by a k-way merge gives the
global result.  No other collective exists on this path.

The exchange lives behind the C ABI (csrc/shard_group.hip), no torch involved:
  ShardGroup  one process, a host thread per shard, records written into the root device's gather
              buffer over peer mappings (mse_shard_group_*); shards may share a device.
  Comm        one process per GPU: one ncclAllGather of the packed records through librccl.so on the
              searcher's stream + merge (mse_comm_*); the 128-byte id reaches the ranks through
              whatever rendezvous the host has (bench.py: a gloo broadcast).
merge_topk_numpy / merge_topk_torch / all_gather_topk are the host-side restatement of the merge
rule, used by the CPU (gloo) tests of the plumbing.
"""
import ctypes as C

import numpy as np

from . import ffi
from .ffi import check, check_ptr
from .vector import MODE_AUTO, Searcher, _bits, _p


class _BorrowedSearcher(Searcher):
    """A shard's searcher, owned by its ShardGroup (timing / certificate statistics only)."""

    def __init__(self, handle):   # noqa: D401 -- no mse_searcher_new: the group made it
        self.vecs = None
        self._h = handle

    def close(self):
        self._h = None


class _BorrowedVectorList:
    """A shard's rows, owned by its ShardGroup: what a codec / graph of the shard is built over."""

    def __init__(self, handle, d):
        self._h = handle
        self.d_emb = d

    def __len__(self):
        return int(ffi.lib().mse_base_len(self._h))

    def rows(self, first, n):
        out = np.empty((n, self.d_emb), np.uint16)
        check(ffi.lib().mse_base_read_rows(self._h, first, n, _p(out, C.c_uint16)), "mse_base_read_rows")
        return out

    def close(self):
        self._h = None


class ShardGroup:
    """Row-sharded index driven by one process: mse_shard_group (include/mse.h)."""

    def __init__(self, n_shards, d=1152, devices=None):
        dev = None
        if devices is not None:
            if len(devices) != n_shards:
                raise ValueError("one device ordinal per shard")
            dev = (C.c_int * n_shards)(*[int(x) for x in devices])
        self.d = d
        self._h = check_ptr(ffi.lib().mse_shard_group_new(dev, n_shards, d), "mse_shard_group_new")

    @property
    def n_shards(self):
        return int(ffi.lib().mse_shard_group_n_shards(self._h))

    def __len__(self):
        return int(ffi.lib().mse_shard_group_len(self._h))

    def device(self, shard):
        return int(ffi.lib().mse_shard_group_device(self._h, shard))

    def peer_mapped(self, shard):
        return bool(ffi.lib().mse_shard_group_peer_mapped(self._h, shard))

    def searcher(self, shard):
        return _BorrowedSearcher(check_ptr(ffi.lib().mse_shard_group_searcher(self._h, shard), "mse_shard_group_searcher"))

    def generate(self, seed, first_row, total_rows):
        check(ffi.lib().mse_shard_group_generate(self._h, seed, first_row, total_rows), "mse_shard_group_generate")

    def load_host(self, f16s):
        a = _bits(f16s).reshape(-1, self.d)
        check(ffi.lib().mse_shard_group_load_host(self._h, _p(a, C.c_uint16), a.shape[0]), "mse_shard_group_load_host")

    def set_shard_device(self, shard, rows_dev, n_rows, first_row):
        check(ffi.lib().mse_shard_group_set_shard_device(self._h, shard, rows_dev, n_rows, first_row), "mse_shard_group_set_shard_device")

    def bruteforce_topk(self, queries, k, mode=MODE_AUTO):
        """Same contract as Searcher.bruteforce_topk over the whole (sharded) index; ids are global."""
        q = _bits(queries).reshape(-1, self.d)
        nq = q.shape[0]
        scores = np.empty((nq, k), np.int64)
        ids = np.empty((nq, k), np.uint32)
        check(ffi.lib().mse_shard_group_search(self._h, _p(q, C.c_uint16), nq, k, mode, _p(scores, C.c_int64), _p(ids, C.c_uint32)),
              "mse_shard_group_search")
        return scores, ids

    def bruteforce_topk_dev(self, queries_dev, nq, k, scores_dev, ids_dev, mode=MODE_AUTO):
        check(ffi.lib().mse_shard_group_search_dev(self._h, queries_dev, nq, k, mode, scores_dev, ids_dev), "mse_shard_group_search_dev")

    # ---- the approximate-search paths over the same shards ----
    def base(self, shard):
        """The shard's rows (a borrowed VectorList): build the shard's codes / graph over it, on the shard's device
        (mse.ffi.lib().mse_set_device(group.device(shard)) first when shards live on several devices)."""
        return _BorrowedVectorList(check_ptr(ffi.lib().mse_shard_group_base(self._h, shard), "mse_shard_group_base"), self.d)

    def first_row(self, shard):
        return int(ffi.lib().mse_shard_group_first_row(self._h, shard))

    def attach_pq(self, shard, quantizer, codes):
        """The shard's codec and the PQ codes (+ descriptor bytes) of ITS rows; None, None detaches.  The group keeps references."""
        check(ffi.lib().mse_shard_group_attach_pq(self._h, shard, quantizer._h if quantizer is not None else None,
                                                  codes._h if codes is not None else None), "mse_shard_group_attach_pq")
        self.__dict__.setdefault("_keep", {})[("pq", shard)] = (quantizer, codes)

    def attach_graph(self, shard, dgraph):
        """A graph over the shard's rows (DeviceGraph / BuildGraph, with its entry table set); None detaches."""
        check(ffi.lib().mse_shard_group_attach_graph(self._h, shard, dgraph._h if dgraph is not None else None), "mse_shard_group_attach_graph")
        self.__dict__.setdefault("_keep", {})[("graph", shard)] = dgraph

    def pq_scan_topk(self, queries_f32, r, k, scales=None):
        """ProductQuantizer.scan_topk_batch over the sharded codes with the unsharded call's answer bit for bit: the index's top-r by
        ADC (exchange 1), exact fp16 re-score of each shard's members (exchange 2), top-k.  Returns (scores, global ids)."""
        q = np.ascontiguousarray(queries_f32, np.float32).reshape(-1, self.d)
        nq = q.shape[0]
        sc = None if scales is None else np.ascontiguousarray(scales, np.float32).reshape(-1)
        scores, ids = np.empty((nq, k), np.int64), np.empty((nq, k), np.uint32)
        check(ffi.lib().mse_shard_group_pq_scan_topk(self._h, _p(q, C.c_float), _p(sc, C.c_float) if sc is not None else None, nq, int(r), int(k),
                                                     _p(scores, C.c_int64), _p(ids, C.c_uint32)), "mse_shard_group_pq_scan_topk")
        return scores, ids

    def query_topk(self, queries, k, luts=None, scales=None, disable_pq=True, beamwidth=4, search_list=64):
        """disk_query_topk on every shard's own graph (its entry table, greedy_search, the k best visited records), one exchange,
        merge by (score desc, id asc).  Returns (scores, global ids)."""
        q = _bits(queries).reshape(-1, self.d)
        nq = q.shape[0]
        tables = None if luts is None else np.ascontiguousarray(luts, np.float32).reshape(nq, -1)
        sc = None
        if scales is not None:
            sc = np.ascontiguousarray(scales, np.float32)
            if sc.ndim == 1:
                sc = np.ascontiguousarray(np.broadcast_to(sc, (nq, sc.size)))
        scores, ids = np.empty((nq, k), np.int64), np.empty((nq, k), np.uint32)
        check(ffi.lib().mse_shard_group_query_topk(self._h, _p(q, C.c_uint16), _p(tables, C.c_float) if tables is not None else None,
                                                   _p(sc, C.c_float) if sc is not None else None, nq, int(bool(disable_pq)), int(beamwidth),
                                                   int(search_list), int(k), _p(scores, C.c_int64), _p(ids, C.c_uint32)), "mse_shard_group_query_topk")
        return scores, ids

    EXCHANGE_PEER, EXCHANGE_RCCL = 0, 1

    def set_exchange(self, kind):
        """How the per-shard records meet: EXCHANGE_PEER (peer stores / staged copies, default) or EXCHANGE_RCCL (one
        ncclAllGather per search; every shard on its own device).  Raises MseError -- and keeps the previous exchange -- when
        RCCL cannot be brought up."""
        check(ffi.lib().mse_shard_group_set_exchange(self._h, int(kind)), "mse_shard_group_set_exchange")

    @property
    def exchange(self):
        return int(ffi.lib().mse_shard_group_exchange(self._h))

    @property
    def rccl_ranks(self):
        return int(ffi.lib().mse_shard_group_rccl_ranks(self._h))

    def last_timing(self):
        """Breakdown of the last search in ms: local search of the slowest shard, its exchange leg, the merge, wall clock."""
        out = (C.c_double * 4)()
        check(ffi.lib().mse_shard_group_last_timing(self._h, out), "mse_shard_group_last_timing")
        return {"local_search_ms": out[0], "exchange_ms": out[1], "merge_ms": out[2], "wall_ms": out[3]}

    def close(self):
        if self._h:
            ffi.lib().mse_shard_group_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Comm:
    """One rank of the one-process-per-GPU layout: mse_comm (RCCL all-gather of packed top-k records + merge)."""

    ID_BYTES = 128

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(Comm.ID_BYTES)
        check(ffi.lib().mse_comm_unique_id(buf), "mse_comm_unique_id")
        return buf.raw

    def __init__(self, unique_id, rank, world):
        if len(unique_id) != Comm.ID_BYTES:
            raise ValueError("the communicator id is 128 bytes")
        buf = C.create_string_buffer(bytes(unique_id), Comm.ID_BYTES)
        self._h = check_ptr(ffi.lib().mse_comm_init(buf, rank, world), "mse_comm_init")

    @property
    def rank(self):
        return int(ffi.lib().mse_comm_rank(self._h))

    @property
    def size(self):
        return int(ffi.lib().mse_comm_size(self._h))

    def search_dev(self, searcher, queries_dev, nq, k, scores_dev, ids_dev, mode=MODE_AUTO, id_offset=0):
        check(ffi.lib().mse_comm_search_dev(self._h, searcher._h, queries_dev, nq, k, mode, id_offset, scores_dev, ids_dev),
              "mse_comm_search_dev")

    def exchange_dev(self, searcher, block_dev, nq, k_in, k, scores_dev, ids_dev):
        """This rank's packed block of (score, global id) records -> one all-gather -> the k best per query (device pointers)."""
        check(ffi.lib().mse_comm_exchange_dev(self._h, searcher._h, block_dev, nq, k_in, k, scores_dev, ids_dev), "mse_comm_exchange_dev")

    def pq_scan_topk(self, quantizer, codes, searcher, queries_f32, r, k, first_row, scores_dev, ids_dev, scales=None):
        """One rank of the sharded PQ scan + exact re-rank (two exchanges); outputs [nq, k] on this rank's device."""
        q = np.ascontiguousarray(queries_f32, np.float32).reshape(-1, quantizer.n_dims)
        sc = None if scales is None else np.ascontiguousarray(scales, np.float32).reshape(-1)
        check(ffi.lib().mse_comm_pq_scan_topk(self._h, quantizer._h, codes._h, searcher._h, _p(q, C.c_float), _p(sc, C.c_float) if sc is not None else None,
                                              q.shape[0], int(r), int(k), int(first_row), scores_dev, ids_dev), "mse_comm_pq_scan_topk")

    def query_topk(self, searcher, dgraph, queries, k, first_row, scores_dev, ids_dev, quantizer=None, codes=None, luts=None, scales=None,
                   disable_pq=True, beamwidth=4, search_list=64):
        """One rank of the sharded graph index: this rank's graph answers the batch, one all-gather, merge; outputs on the device."""
        if isinstance(queries, tuple):
            q_ptr, nq = C.cast(C.c_void_p(int(queries[0])), C.POINTER(C.c_uint16)), int(queries[1])
        else:
            q = _bits(queries)
            q = q.reshape(-1, q.shape[-1])
            nq, q_ptr = q.shape[0], _p(q, C.c_uint16)
        tables = None if luts is None else np.ascontiguousarray(luts, np.float32).reshape(nq, -1)
        sc = None if scales is None else np.ascontiguousarray(scales, np.float32)
        check(ffi.lib().mse_comm_query_topk(self._h, searcher._h, quantizer._h if quantizer is not None else None, codes._h if codes is not None else None,
                                            dgraph._h, q_ptr, _p(tables, C.c_float) if tables is not None else None,
                                            _p(sc, C.c_float) if sc is not None else None, nq, int(bool(disable_pq)), int(beamwidth), int(search_list),
                                            int(k), int(first_row), scores_dev, ids_dev), "mse_comm_query_topk")

    def last_timing(self):
        """This rank's last search_dev in ms (waits for it): local search, all-gather (incl. waiting for the slowest rank), merge."""
        out = (C.c_double * 4)()
        check(ffi.lib().mse_comm_last_timing(self._h, out), "mse_comm_last_timing")
        return {"local_search_ms": out[0], "exchange_ms": out[1], "merge_ms": out[2], "sum_ms": out[3]}

    def close(self):
        if self._h:
            ffi.lib().mse_comm_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


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


def two_phase_pq_scan(local_adc_topr, local_exact, lo, hi, r, k, group=None):
    """Host restatement of the sharded PQ scan's protocol (csrc/shard_group.hip: mse_shard_group_pq_scan_topk / mse_comm_pq_scan_topk),
    for the CPU (gloo) tests of the plumbing.  local_adc_topr() -> (scores int64 [nq, r], LOCAL ids uint32 [nq, r]) of this rank's
    codes (ID_NONE = empty); local_exact(q, local_id) -> exact i64 score of this rank's row.  Rows [lo, hi) live here.
      A  every rank's ADC top-r with global ids -> all-gather -> the index's top-r by (ADC score desc, id asc)
      B  every rank scores ITS members of that list exactly, the others' slots stay empty -> all-gather -> top-k by (score desc, id asc)
    Returns (scores [nq, k] int64, ids [nq, k] int64 holding u32 values), identical on every rank."""
    import torch
    s, i = local_adc_topr()
    gid = np.where(i == 0xFFFFFFFF, np.int64(0xFFFFFFFF), i.astype(np.int64) + lo)
    _, top_ids = all_gather_topk(torch.from_numpy(np.ascontiguousarray(s, np.int64)), torch.from_numpy(gid), r, group)
    top_ids = top_ids.numpy()
    nq = top_ids.shape[0]
    ex_s = np.full((nq, r), np.iinfo(np.int64).min, np.int64)
    ex_i = np.full((nq, r), 0xFFFFFFFF, np.int64)
    for q in range(nq):
        for j in range(r):
            g = int(top_ids[q, j])
            if g != 0xFFFFFFFF and lo <= g < hi:
                ex_s[q, j] = local_exact(q, g - lo)
                ex_i[q, j] = g
    return all_gather_topk(torch.from_numpy(ex_s), torch.from_numpy(ex_i), k, group)
