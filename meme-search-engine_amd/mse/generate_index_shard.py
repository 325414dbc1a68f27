"""generate-index-shard (src/generate_index_shard.rs) over the device Vamana build.

Reads one shard file written by dump-processor (`{i}.shard.msgpack`: a ShardInputHeader followed by ShardedRecords,
src/common.rs:131-142, src/dump_processor.rs:203-207,447-455), optionally appends the query vectors of the
OOD-DiskANN variant (`queries.bin`, raw fp16 rows, generate_index_shard.rs:73-83), builds the graph on the GPU in the
reference's order of passes (random fill, first pass, optional second pass with alpha_2, robust stitch; :102-133) and
writes `{id}.shard.bin` + `{id}.shard-header.msgpack` exactly as the reference lays them out (:139-164).

rmp-serde's default struct encoding is positional (a msgpack array of the fields in declaration order; only
`to_vec_named` writes maps), `serde_bytes` fields are msgpack bin, f32 is msgpack float32 -- these files follow that.

The reference seeds fastrand from the clock (:46); here `seed` names the shuffle and the random initial graph, so a run
can be repeated.  There is no CPU path: every score comes from libmse_hip.so.
"""
import argparse
import os

import msgpack
import numpy as np

from . import diskann
from .vector import VectorList, Searcher

D_EMB = 1152   # generate_index_shard.rs:40


def read_shard_input(path, d_emb=D_EMB):
    """-> (header dict {id, centroid}, original_ids uint32 [n], vectors uint16 [n, d_emb])   (:48-69)"""
    ids, chunks = [], []
    with open(path, "rb") as f:
        unp = msgpack.Unpacker(f, raw=True, max_buffer_size=0)
        try:
            hid, centroid = next(unp)
        except StopIteration:
            raise ValueError("shard file has no header")
        for rec in unp:
            rid, vec = rec
            if len(vec) != d_emb * 2:
                raise ValueError("record vector is not %d fp16 values" % d_emb)
            ids.append(rid)
            chunks.append(vec)
    vecs = np.frombuffer(b"".join(chunks), dtype="<u2").reshape(len(ids), d_emb) if ids else np.empty((0, d_emb), np.uint16)
    return {"id": int(hid), "centroid": [float(x) for x in centroid]}, np.asarray(ids, np.uint32), vecs


def write_shard_input(path, shard_id, centroid, ids, vectors):
    """What dump-processor writes for one shard (dump_processor.rs:203-207,447-455); used by tests and tools."""
    v = np.ascontiguousarray(vectors).view(np.uint16).reshape(len(ids), -1)
    with open(path, "wb") as f:
        f.write(msgpack.packb([int(shard_id), [float(x) for x in centroid]], use_single_float=True))
        for i, rid in enumerate(ids):
            f.write(msgpack.packb([int(rid), v[i].astype("<u2").tobytes()], use_bin_type=True))


def write_shard_output(out_dir, header, medioid, original_ids, graph, query_breakpoint):
    """`{id}.shard.bin` = the base nodes' lists back to back as u32 LE; header = ShardHeader (common.rs:144-152) with
    byte offsets (one extra entry at the end, :153).  Returns the two paths."""
    deg = graph.deg[:query_breakpoint].astype(np.int64)
    offsets = np.concatenate([[0], np.cumsum(deg * 4)])
    bin_path = os.path.join(out_dir, "%d.shard.bin" % header["id"])
    with open(bin_path, "wb") as f:
        mask = np.arange(graph.adj.shape[1])[None, :] < deg[:, None]
        f.write(graph.adj[:query_breakpoint][mask].astype("<u4").tobytes())
    hdr_path = os.path.join(out_dir, "%d.shard-header.msgpack" % header["id"])
    with open(hdr_path, "wb") as f:
        f.write(msgpack.packb([header["id"], int(original_ids.max()), header["centroid"], int(medioid),
                               [int(o) for o in offsets], [int(i) for i in original_ids]], use_single_float=True))
    return bin_path, hdr_path


def read_shard_output(out_dir, shard_id):
    """-> (ShardHeader as dict, list of neighbour arrays); the reader dump-processor applies (dump_processor.rs:239)."""
    with open(os.path.join(out_dir, "%d.shard-header.msgpack" % shard_id), "rb") as f:
        sid, mx, centroid, medioid, offsets, mapping = msgpack.unpackb(f.read(), raw=True)
    data = np.fromfile(os.path.join(out_dir, "%d.shard.bin" % shard_id), dtype="<u4")
    lists = [data[offsets[i] // 4:offsets[i + 1] // 4] for i in range(len(mapping))]
    return {"id": sid, "max": mx, "centroid": centroid, "medioid": medioid, "offsets": offsets, "mapping": mapping}, lists


def generate_index_shard(input_file, out_dir, queries_bin=None, l=192, r=64, maxc=750, alpha=65536, query_alpha=65536,
                         alpha_2=65536, second_pass=False, seed=None, batch=2048, d_emb=D_EMB, log=print):
    """main() of generate_index_shard.rs (:43-169).  Returns (ShardHeader fields, host IndexGraph)."""
    header, original_ids, vecs = read_shard_input(input_file, d_emb)
    if len(original_ids) == 0:
        raise ValueError("shard file holds no records")          # the reference panics at :158 (max of nothing)
    query_breakpoint = len(original_ids)                         # :71
    if queries_bin:
        q = np.fromfile(queries_bin, dtype="<u2")
        vecs = np.concatenate([vecs, q[:q.size - q.size % d_emb].reshape(-1, d_emb)])   # :73-83
    n = len(vecs)
    rng = np.random.default_rng(seed)
    base = VectorList.from_f16s(vecs, d_emb)
    searcher = Searcher(base)
    graph = diskann.BuildGraph(n, r)                             # IndexGraph::empty (:102)
    graph.random_fill(int(rng.integers(0, 2 ** 32)))             # :104-107
    medioid = diskann.medioid(base)                              # :111
    cfg = dict(r=r, l=l, maxc=maxc, alpha=alpha, query_alpha=query_alpha, saturate_graph=False,
               query_breakpoint=query_breakpoint, max_add_per_stitch_iter=16)   # :85-94
    graph.build(searcher, rng.permutation(n).astype(np.uint32), medioid, diskann.IndexBuildConfig(**cfg), batch)   # :113-116
    if second_pass:                                              # :120-127
        cfg["alpha"] = alpha_2
        graph.build(searcher, rng.permutation(n).astype(np.uint32), medioid, diskann.IndexBuildConfig(**cfg), batch)
    if query_breakpoint < n:                                     # :129-133
        qorder = (query_breakpoint + rng.permutation(n - query_breakpoint)).astype(np.uint32)
        graph.robust_stitch(searcher, qorder, diskann.IndexBuildConfig(**cfg))
    host = graph.to_host()
    deg = host.deg[:query_breakpoint]
    log("average degree %.3f, min %d, max %d" % (deg.mean(), deg.min(), deg.max()))   # report_degrees (lib.rs:398-411)
    write_shard_output(out_dir, header, medioid, original_ids, host, query_breakpoint)
    log("%d vectors" % query_breakpoint)                         # :166
    return {"id": header["id"], "medioid": medioid, "query_breakpoint": query_breakpoint}, host


def main(argv=None):
    ap = argparse.ArgumentParser(description="Generate indices from shard files")   # same flags as the reference (:13-38)
    ap.add_argument("input_file")
    ap.add_argument("out_dir")
    ap.add_argument("queries_bin", nargs="?")
    ap.add_argument("-L", dest="l", type=int, default=192, help="search list size (higher is better but slower)")
    ap.add_argument("-R", dest="r", type=int, default=64, help="graph degree")
    ap.add_argument("-C", dest="maxc", type=int, default=750, help="max candidate list size")
    ap.add_argument("-A", dest="alpha", type=int, default=65536, help="first pass relaxation factor (times 2^16)")
    ap.add_argument("-Q", dest="query_alpha", type=int, default=65536, help="query set special relaxation factor (times 2^16)")
    ap.add_argument("-B", dest="alpha_2", type=int, default=65536, help="second pass relaxation factor (times 2^16)")
    ap.add_argument("-s", dest="second_pass", action="store_true", help="do second pass")
    ap.add_argument("-N", dest="n", type=int, default=None, help="number of vectors to allocate for (accepted, unused)")
    ap.add_argument("--seed", type=int, default=None, help="names the shuffles and the random initial graph (the reference uses the clock)")
    ap.add_argument("--batch", type=int, default=2048, help="points inserted per device batch (1 = the reference's sequential loop)")
    a = ap.parse_args(argv)
    generate_index_shard(a.input_file, a.out_dir, a.queries_bin, a.l, a.r, a.maxc, a.alpha, a.query_alpha, a.alpha_2,
                         a.second_pass, a.seed, a.batch)


if __name__ == "__main__":
    main()


# ---- merging the shard graphs (src/dump_processor.rs:219-296) ------------------------------------------------------------

SHARD_SPILL = 2   # dump_processor.rs:136: every record is written to its two closest shards


def merge_shards(shards_dir, shard_ids=None, spill=SHARD_SPILL):
    """What dump-processor does with the outputs of generate-index-shard before it packs index.bin: for every original id, the
    union of its neighbour lists from the (at most `spill`) shards that hold it, within-shard ids mapped back to original ids,
    first occurrence kept (`!out_vertices.contains`, :286-289).  The reference fills a record's shard slots in directory-listing
    order (:225,248-254), which the OS decides; here shards are taken in ascending id.
    -> (adj uint32 [n][spill * max list], deg uint32 [n], shards_of int32 [n][spill] (-1 = none),
        shard_specs [(centroid float32 [d], medioid as ORIGINAL id)] in ascending shard id (IndexHeader.shards, :261))."""
    if shard_ids is None:
        shard_ids = sorted(int(f.split(".")[0]) for f in os.listdir(shards_dir) if f.endswith(".shard-header.msgpack"))
    none = np.uint32(0xFFFFFFFF)
    loaded, n, widest = [], 0, 0
    for sid in shard_ids:
        h, ls = read_shard_output(shards_dir, sid)
        loaded.append((sid, h, ls))
        n = max(n, int(h["max"]) + 1)
        widest = max(widest, max((len(l) for l in ls), default=0))
    width = max(spill * widest, 1)
    slots = np.full((n, width), none, np.uint32)      # slot block c of a record = its list in the c-th shard that holds it
    shards_of = np.full((n, spill), -1, np.int32)
    filled = np.zeros(n, np.int64)
    specs = []
    for sid, h, ls in loaded:
        mapping = np.asarray(h["mapping"], np.uint32)
        specs.append((np.asarray(h["centroid"], np.float32), int(mapping[h["medioid"]])))
        # :245-259: every occurrence of an id takes the record's next empty slot -- also when one shard's mapping repeats the id
        # (occurrence r of an id inside this shard goes to slot filled + r); no empty slot left = "shard processing inconsistency"
        by_id = np.argsort(mapping, kind="stable")
        sorted_ids = mapping[by_id]
        first = np.r_[True, sorted_ids[1:] != sorted_ids[:-1]] if len(mapping) else np.zeros(0, bool)
        run_start = np.maximum.accumulate(np.where(first, np.arange(len(mapping)), 0)) if len(mapping) else np.zeros(0, np.int64)
        occ = np.empty(len(mapping), np.int64)
        occ[by_id] = np.arange(len(mapping)) - run_start
        if ((filled[mapping] + occ) >= spill).any():
            raise ValueError("shard processing inconsistency")
        lens = np.fromiter((len(l) for l in ls), np.int64, len(ls))
        local = np.full((len(ls), widest), 0, np.int64)
        mask = np.arange(widest)[None, :] < lens[:, None]
        if mask.any():
            local[mask] = np.concatenate(ls).astype(np.int64)
        glob = np.where(mask, mapping[local], none)                         # within-shard ids -> original ids
        c = filled[mapping] + occ
        for k in range(spill):
            pick = c == k
            slots[mapping[pick], k * widest:(k + 1) * widest] = glob[pick]
            shards_of[mapping[pick], k] = sid
        np.add.at(filled, mapping, 1)
    # first occurrence kept, order kept (`!out_vertices.contains`), a few thousand records at a time
    # The slot array is compacted in place and returned as the adjacency array, so the peak is ONE dense n x width u32 array
    # (the reference streams per record, :264-293; at 1e8 records and R = 64 this array is 51 GB -- merge in id ranges via
    # `shard_ids` subsets and np.memmap outputs beyond that).
    deg = np.zeros(n, np.uint32)
    later = np.arange(width)[:, None] > np.arange(width)[None, :]
    for i in range(0, n, 4096):
        blk = slots[i:i + 4096]
        dup = ((blk[:, :, None] == blk[:, None, :]) & later[None]).any(axis=2)
        keep = (blk != none) & ~dup
        order = np.argsort(~keep, axis=1, kind="stable")
        slots[i:i + 4096] = np.where(np.take_along_axis(keep, order, axis=1), np.take_along_axis(blk, order, axis=1), 0)
        deg[i:i + 4096] = keep.sum(axis=1)
    return slots, deg, shards_of, specs
