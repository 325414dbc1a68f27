"""The on-disk index of query-disk-index (src/query_disk_index.rs:658-709, written by src/dump_processor.rs:306-313,
463-569) as far as its formats are pinned by the reference's own sources:

  index.msgpack                rmp-serde `to_vec_named` of IndexHeader (src/common.rs:166-174): a msgpack MAP with keys
                               shards [[centroid f32 x d, medioid u32], ...], count, dead_count, record_pad_size,
                               quantizer {centroids, transform, n_dims_per_code, n_dims} (diskann/src/vector.rs:308-314)
                               and descriptor_cdfs [[f32, ...], ...]
  index.pq-codes.bin           count x (n_dims / n_dims_per_code) bytes, row major (query_disk_index.rs:101-104,678)
  index.descriptor-codes.bin   count x len(descriptor_cdfs) bytes (:133-142,679)
  index.bin                    count x record_pad_size bytes; record = [u16 LE length][payload][zero padding]
                               (:73-81; dump_processor.rs:505-521)

The payload of an index.bin record is `bitcode::encode(PackedIndexEntry)` (bitcode 0.6.7, src/common.rs:154-164).  bitcode's
wire format is not described anywhere in the reference tree and the crate's source is not available here: mse/bitcode06.py
restates it for this struct from knowledge of the crate (parity UNPINNED -- see that module), encoder and decoder written
together.  Because no byte string produced by the real crate has been checked against it, it is NOT a default: `DiskIndex` and
`write_index` take the payload codec as an explicit argument (`UNPINNED_BITCODE06_DECODE` / `UNPINNED_BITCODE06_ENCODE` to opt
in to the restatement, or a codec backed by the real crate), and a directory written with the restatement is not claimed to be
readable by the reference until one golden `PackedIndexEntry` from the crate is a test fixture.
`DiskIndex.to_device()` is the front door of the GPU-resident beam search: vectors, adjacency (+ the "has a URL" flags the
search filters on, query_disk_index.rs:172), PQ codes and descriptor bytes go to HBM in one step; `write_index` is what
dump-processor's packing loop writes (dump_processor.rs:463-569) so that the build side of this package produces a directory
the read side (and, format permitting, the reference) opens.
"""
import os

import msgpack
import numpy as np

from . import bitcode06
from .vector import ProductQuantizer, Codes

RECORD_PAD_SIZE = 4096   # dump_processor.rs:135

# opt-in payload codecs (restated bitcode 0.6, parity UNPINNED: never a silent default)
UNPINNED_BITCODE06_DECODE = bitcode06.decode_packed_index_entry
UNPINNED_BITCODE06_ENCODE = bitcode06.encode_packed_index_entry


class IndexHeader:
    """src/common.rs:166-174."""

    def __init__(self, shards, count, dead_count, record_pad_size, quantizer, descriptor_cdfs):
        self.shards = shards                      # list of (centroid float32 [d], medioid)
        self.count = int(count)
        self.dead_count = int(dead_count)
        self.record_pad_size = int(record_pad_size)
        self.quantizer = quantizer                # dict: centroids, transform (float32 arrays), n_dims_per_code, n_dims
        self.descriptor_cdfs = descriptor_cdfs    # list of float32 arrays

    @property
    def pq_code_size(self):                       # query_disk_index.rs:678
        return self.quantizer["n_dims"] // self.quantizer["n_dims_per_code"]

    @property
    def n_descriptors(self):                      # :679
        return len(self.descriptor_cdfs)

    def product_quantizer(self) -> ProductQuantizer:
        q = self.quantizer
        d = q["n_dims"]
        return ProductQuantizer(q["centroids"].reshape(-1, d), q["transform"].reshape(d, d), q["n_dims_per_code"], d)

    def shard_centroids(self):
        return np.stack([c for c, _ in self.shards]).astype(np.float32)


def read_index_header(path) -> IndexHeader:
    with open(path, "rb") as f:
        m = msgpack.unpackb(f.read(), raw=False, strict_map_key=False)
    need = {"shards", "count", "dead_count", "record_pad_size", "quantizer", "descriptor_cdfs"}
    if not isinstance(m, dict) or not need <= set(m):
        raise ValueError("index.msgpack is not an IndexHeader map")
    q = m["quantizer"]
    quant = {"centroids": np.asarray(q["centroids"], np.float32), "transform": np.asarray(q["transform"], np.float32),
             "n_dims_per_code": int(q["n_dims_per_code"]), "n_dims": int(q["n_dims"])}
    d = quant["n_dims"]
    if quant["transform"].size != d * d or quant["centroids"].size % d or d % quant["n_dims_per_code"]:
        raise ValueError("index.msgpack: quantizer arrays do not match n_dims")   # the asserts of vector.rs:334-337
    shards = [(np.asarray(c, np.float32), int(med)) for c, med in m["shards"]]
    return IndexHeader(shards, m["count"], m["dead_count"], m["record_pad_size"], quant,
                       [np.asarray(c, np.float32) for c in m["descriptor_cdfs"]])


def write_index_header(path, header: IndexHeader):
    """`rmp_serde::to_vec_named(&header)` (dump_processor.rs:559-568): field order of the struct, f32 as float32."""
    q = header.quantizer
    m = {"shards": [[[float(x) for x in c], int(med)] for c, med in header.shards],
         "count": header.count, "dead_count": header.dead_count, "record_pad_size": header.record_pad_size,
         "quantizer": {"centroids": [float(x) for x in np.asarray(q["centroids"]).reshape(-1)],
                       "transform": [float(x) for x in np.asarray(q["transform"]).reshape(-1)],
                       "n_dims_per_code": int(q["n_dims_per_code"]), "n_dims": int(q["n_dims"])},
         "descriptor_cdfs": [[float(x) for x in c] for c in header.descriptor_cdfs]}
    with open(path, "wb") as f:
        f.write(msgpack.packb(m, use_single_float=True))


class DiskIndex:
    """An index directory opened the way initialize_index / initialize_memory_maps do (query_disk_index.rs:658-709)."""

    def __init__(self, path, decode_entry=None):
        self.path = path
        self.header = read_index_header(os.path.join(path, "index.msgpack"))
        h = self.header
        self.pq_codes = np.memmap(os.path.join(path, "index.pq-codes.bin"), dtype=np.uint8, mode="r")
        self.descriptors = np.memmap(os.path.join(path, "index.descriptor-codes.bin"), dtype=np.uint8, mode="r")
        if self.pq_codes.size != h.count * h.pq_code_size:
            raise ValueError("index.pq-codes.bin does not hold count x pq_code_size bytes")
        if self.descriptors.size != h.count * h.n_descriptors:
            raise ValueError("index.descriptor-codes.bin does not hold count x n_descriptors bytes")
        self.pq_codes = self.pq_codes.reshape(h.count, h.pq_code_size)
        self.descriptors = self.descriptors.reshape(h.count, max(h.n_descriptors, 1)) if h.n_descriptors else None
        if decode_entry == "unpinned-bitcode06":      # the same opt-in, spelled as a string (configuration files)
            decode_entry = UNPINNED_BITCODE06_DECODE
        if decode_entry is not None and not callable(decode_entry):
            raise ValueError("decode_entry must be a callable or 'unpinned-bitcode06'")
        self.decode_entry = decode_entry
        self._data = os.path.join(path, "index.bin")

    def device_codes(self) -> Codes:
        """PQ codes + descriptor bytes resident in HBM (the reference keeps them in locked mmaps, :686-707)."""
        return Codes(np.ascontiguousarray(self.pq_codes), None if self.descriptors is None else np.ascontiguousarray(self.descriptors))

    def record_payload(self, idx) -> bytes:
        """read_node (:73-81) up to the bitcode call: the payload bytes of record `idx`."""
        pad = self.header.record_pad_size
        with open(self._data, "rb") as f:
            f.seek(idx * pad)
            buf = f.read(pad)
        if len(buf) != pad:
            raise ValueError("index.bin ends inside record %d" % idx)
        n = int.from_bytes(buf[:2], "little")
        if n + 2 > pad:
            raise ValueError("record %d: length prefix exceeds the record" % idx)
        return buf[2:2 + n]

    def read_node(self, idx):
        """read_node (:73-81): the PackedIndexEntry of record `idx` (a dict with the struct's field names)."""
        if self.decode_entry is None:
            raise NotImplementedError("no decode_entry: index.bin payloads are bitcode-encoded PackedIndexEntry; pass "
                                      "decode_entry=UNPINNED_BITCODE06_DECODE to opt in to the unpinned restatement (mse/bitcode06.py)")
        return self.decode_entry(self.record_payload(idx))

    def entries(self):
        """All records in id order, read sequentially (one pass over index.bin)."""
        if self.decode_entry is None:
            raise NotImplementedError("no decode_entry: pass decode_entry=UNPINNED_BITCODE06_DECODE to opt in to the unpinned restatement")
        pad, h = self.header.record_pad_size, self.header
        with open(self._data, "rb") as f:
            for idx in range(h.count):
                buf = f.read(pad)
                if len(buf) != pad:
                    raise ValueError("index.bin ends inside record %d" % idx)
                n = int.from_bytes(buf[:2], "little")
                if n + 2 > pad:
                    raise ValueError("record %d: length prefix exceeds the record" % idx)
                yield self.decode_entry(buf[2:2 + n])

    def to_device(self):
        """Everything the GPU-resident beam search needs, resident in HBM:
        -> (VectorList of the fp16 vectors, DeviceGraph with the has-url flags, Codes, IndexGraph host copy, urls list).
        Records whose URL is empty are graph-only nodes (dump_processor.rs:510-517): traversed, never returned (:172)."""
        from .vector import VectorList
        from .diskann import DeviceGraph, IndexGraph
        if self.decode_entry is None:     # say so before any work is done, not at the first record
            raise NotImplementedError("DiskIndex.to_device needs a record codec: open the directory with "
                                      "decode_entry=UNPINNED_BITCODE06_DECODE (or 'unpinned-bitcode06') to opt in to the unpinned restatement")
        h = self.header
        d = h.quantizer["n_dims"]
        vecs = np.zeros((h.count, d), np.uint16)
        lists, urls = [], []
        for i, e in enumerate(self.entries()):
            if e["id"] != i:
                raise ValueError("record %d carries id %d" % (i, e["id"]))          # ids are positions (dump_processor.rs:501)
            if len(e["vector"]) != d:
                raise ValueError("record %d: vector has %d components, the quantiser %d" % (i, len(e["vector"]), d))
            vecs[i] = e["vector"]
            lists.append(np.asarray(e["vertices"], np.uint32))
            urls.append(e["url"])
        width = max((len(l) for l in lists), default=0) or 1
        adj = np.zeros((h.count, width), np.uint32)
        deg = np.zeros(h.count, np.uint32)
        for i, l in enumerate(lists):
            adj[i, :len(l)] = l
            deg[i] = len(l)
        if h.count and int(adj.max()) >= h.count:
            raise ValueError("a neighbour list points outside the index")
        has_url = np.array([1 if u else 0 for u in urls], np.uint8)
        graph = IndexGraph(adj, deg)
        return VectorList.from_f16s(vecs, d), DeviceGraph(graph, has_url), self.device_codes(), graph, urls


def write_records(path, payloads, record_pad_size=RECORD_PAD_SIZE):
    """dump_processor.rs:505-521: [u16 LE length][payload] zero-padded to the record size (used by tests and tools)."""
    with open(path, "wb") as f:
        for p in payloads:
            if len(p) > record_pad_size - 2:
                raise ValueError("payload does not fit a record")   # the reference drops such entries (:512)
            f.write(len(p).to_bytes(2, "little") + p + bytes(record_pad_size - 2 - len(p)))


def write_index(out_dir, header: IndexHeader, entries, pq_codes, descriptor_codes, encode_entry=None):
    """The files dump-processor's packing loop leaves behind (dump_processor.rs:306-313,463-569): index.bin (one padded record
    per entry; an entry whose payload does not fit loses its URL and is counted dead, :510-517), index.pq-codes.bin,
    index.descriptor-codes.bin and index.msgpack (count / dead_count filled in here).  `entries` yields PackedIndexEntry dicts
    in id order.  Returns the header as written."""
    if encode_entry is None:
        raise NotImplementedError("no encode_entry: pass UNPINNED_BITCODE06_ENCODE to opt in to the unpinned bitcode 0.6 restatement "
                                  "(its output is not known to be readable by the reference), or an encoder backed by the real crate")
    pad = header.record_pad_size
    count = dead = 0
    with open(os.path.join(out_dir, "index.bin"), "wb") as f:
        for e in entries:
            e = dict(e, id=count)
            payload = encode_entry(e)
            if len(payload) > pad - 2:
                e["url"] = ""
                payload = encode_entry(e)
                dead += 1
                if len(payload) > pad - 2:
                    raise ValueError("record %d does not fit %d bytes even without its URL" % (count, pad))
            f.write(len(payload).to_bytes(2, "little") + payload + bytes(pad - 2 - len(payload)))
            count += 1
    pq_codes = np.ascontiguousarray(pq_codes, np.uint8).reshape(count, -1)
    if pq_codes.shape[1] != header.pq_code_size:
        raise ValueError("pq_codes rows do not match the quantiser")
    pq_codes.tofile(os.path.join(out_dir, "index.pq-codes.bin"))
    dc = np.zeros((count, 0), np.uint8) if descriptor_codes is None else np.ascontiguousarray(descriptor_codes, np.uint8).reshape(count, -1)
    if dc.shape[1] != header.n_descriptors:
        raise ValueError("descriptor rows do not match descriptor_cdfs")
    dc.tofile(os.path.join(out_dir, "index.descriptor-codes.bin"))
    header.count, header.dead_count = count, header.dead_count + dead
    write_index_header(os.path.join(out_dir, "index.msgpack"), header)
    return header
