#!/usr/bin/env python3
"""bench.py -- brute-force dot-product top-k over the 1152-d fp16 index (BASELINE.json metric
"queries/sec over 1e8 x 1152 index", recall@10 = 1.0 because the search is exact).

A "step" = one batch of Q queries scored against the whole index (all shards), top-k returned.
  N = 1 : the whole index lives in one MI355X's HBM (1e8 x 1152 fp16 = 230.4 GB of 288 GB).
  N > 1 : rows are partitioned contiguously over the GPUs (strong scaling, fixed total index); each shard is scanned
          on its GPU, the per-shard [Q, k] records meet in ONE exchange and are merged k-way -- all behind the C ABI
          (csrc/shard_group.hip).  Launched by torchrun (one process per GPU) the exchange is one ncclAllGather through
          librccl; launched bare (`python bench.py --gpus N`) this process drives all N devices with a host thread per
          shard and the records are written into the root device's buffer over peer mappings.
Inputs (index rows and query batches) are synthetic, generated on the device, and resident in HBM
before the timed region.  One JSON line on stdout (rank 0).
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "meme-search-engine_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

D = 1152
SEED_BASE, SEED_QUERY = 0x5EED0001, 0x5EED0002
T_START = 0.0
PMC_TRAFFIC = next((p for p in (os.path.join(ROOT, "profiles", f) for f in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json")) if os.path.exists(p)), "")
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


class _stdout_to_stderr:
    """librccl prints a version banner on fd 1 when a communicator comes up; stdout is reserved for the one JSON line."""

    @staticmethod
    def _flush_c():
        # the banner goes through the C library's buffered stdout: when fd 1 is a file or a pipe it would otherwise be written
        # at exit -- after fd 1 has been restored, behind the JSON line
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:  # noqa: BLE001
            pass

    def __enter__(self):
        sys.stdout.flush()
        self._flush_c()
        self._saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        self._flush_c()
        os.dup2(self._saved, 1)
        os.close(self._saved)


def cpu_baseline(n_rows, k, min_seconds=5.0):
    """The oracle (C restatement of the reference's AVX2 `fast_dot` loop, oracle/mse_oracle.c; kind "port") on the host cores,
    as SURVEY 8(d) specifies: MEASURED at N = 1e5 (BASELINE configs[0], 1000 distinct queries) and at N = 1e7 resident in DRAM
    (23 GB, configs[2]'s size), every mode timed for at least `min_seconds`, nothing extrapolated in `measured`.  Modes:
      fair               every thread scores its own queries against all rows, heap top-k: the thread-per-core shape of the
                         reference's query server (src/query_disk_index.rs:720)
      reference_faithful ONE thread, one query at a time: every row scored, the whole score list ranked, top-k read off the front
                         -- what `evaluate` does (src/query_disk_index.rs:225,262-273)
      index_search       (1e5 only) the in-memory index surface: fp32 queries against fp16 codes, FAISS SQfp16-IP arithmetic in its
                         stated order + heap (src/main.rs:900), thread per request
    `value` is in the metric's unit (queries/s over the bench's row count): the fair rate measured at 1e7 rows scaled by the row
    ratio (x10 for 1e8) -- both sizes stream from DRAM, the scan is linear in rows -- and labelled as scaled."""
    import numpy as np
    from oracle import orc
    orc.build()
    visible = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:     # a container may see every core of the host but own only a slice of their time (cgroup v2 cpu.max = "quota period")
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = max(1, int(round(int(q) / int(per))))
    except Exception:  # noqa: BLE001
        pass
    # threads: the cores this process can actually use.  Under a CPU-time quota the kernel throttles the whole group once the
    # quota of a period is spent, so the best thread count is not the quota: twice the quota measured best on the GPU box
    # (16-core quota on a 256-thread host: 8 / 32 / 64 / 256 threads -> 77 / 100 / 58 / 20 queries/s on a 1e6-row sample)
    cores = min(visible, 2 * quota) if quota else visible
    n_q = 1000
    queries = orc.gen_rows_f16(SEED_QUERY, 0, n_q)

    def run_threads(fn):
        """fn(t, deadline) -> completed queries; all threads start together; returns (total completed, wall seconds)."""
        done = [0] * cores
        t_end = [0.0]

        def body(t):
            done[t] = fn(t, t_end[0])

        ths = [threading.Thread(target=body, args=(t,)) for t in range(cores)]
        t0 = time.perf_counter()
        t_end[0] = t0 + min_seconds
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        return sum(done), time.perf_counter() - t0

    def fair_on(base):
        def fn(t, deadline):          # thread t: queries t, t + cores, ... (cyclic over the 1000) until the time is up
            c, j = 0, t
            while True:
                orc.bruteforce_topk(base, queries[j % n_q:j % n_q + 1], k)
                c += 1
                j += cores
                if time.perf_counter() >= deadline:
                    return c
        n, dt = run_threads(fn)
        return {"queries": n, "seconds": dt, "queries_per_s": n / dt, "threads": cores, "scan_GBps": n / dt * base.shape[0] * D * 2 / 1e9}

    def faithful_on(base):
        t0, c = time.perf_counter(), 0
        while True:
            sc = orc.score_all(base, queries[c % n_q])
            ranks = orc.ranks_from_scores(sc)
            np.flatnonzero(ranks < k)
            c += 1
            if time.perf_counter() - t0 >= min_seconds:
                break
        dt = time.perf_counter() - t0
        return {"queries": c, "seconds": dt, "queries_per_s": c / dt, "threads": 1, "scan_GBps": c / dt * base.shape[0] * D * 2 / 1e9}

    measured = {}
    # ---- N = 1e5: BASELINE configs[0] ----
    block_rows = 100_000
    block = orc.gen_rows_f16(SEED_BASE, 0, block_rows)
    orc.bruteforce_topk(block[:1000], queries[:1], k)  # warm
    m5 = {"rows": block_rows, "distinct_queries": n_q, "fair": fair_on(block), "reference_faithful": faithful_on(block)}
    q32 = orc.f16_to_f32(queries).astype(np.float32) * np.float32(3.0)      # un-normalised fp32 queries (src/common.rs:215-274)

    def idx_fn(t, deadline):
        c, j = 0, t
        while True:
            orc.index_search(block, q32[j % n_q:j % n_q + 1], k, 0)
            c += 1
            j += cores
            if time.perf_counter() >= deadline:
                return c
    n_i, dt_i = run_threads(idx_fn)
    m5["index_search"] = {"queries": n_i, "seconds": dt_i, "queries_per_s": n_i / dt_i, "threads": cores,
                          "what": "orc.index_search: fp32 query x fp16 codes, FAISS SQfp16-IP order 0 + heap, one request per thread"}
    measured["n_1e5"] = m5
    # ---- N = 1e7 resident in DRAM (23 GB), if the host has it ----
    big_rows = 10_000_000
    avail = None
    try:
        avail = next(int(l.split()[1]) * 1024 for l in open("/proc/meminfo") if l.startswith("MemAvailable"))
    except Exception:  # noqa: BLE001
        pass
    need = big_rows * D * 2
    if avail is not None and avail < need + (12 << 30):
        big_rows = max(1_000_000, int((avail - (12 << 30)) // (D * 2) // block_rows * block_rows)) if avail > (14 << 30) else 1_000_000
    # filled by all threads at once so that first touch spreads the pages over every memory controller (one thread's np.tile
    # would put everything on its own NUMA node and the scan would run at one node's bandwidth)
    base = np.empty((big_rows, D), np.uint16)
    chunk = (big_rows + cores - 1) // cores

    def fill(t):     # contiguous np.copyto = memcpy with the GIL released: the fills really run side by side
        r, hi = t * chunk, min(big_rows, (t + 1) * chunk)
        while r < hi:
            off = r % block_rows
            m = min(hi - r, block_rows - off)
            np.copyto(base[r:r + m], block[off:off + m])
            r += m

    t_fill = time.perf_counter()
    fillers = [threading.Thread(target=fill, args=(t,)) for t in range(cores)]
    for th in fillers:
        th.start()
    for th in fillers:
        th.join()
    t_fill = time.perf_counter() - t_fill
    key = "n_1e7" if big_rows == 10_000_000 else f"n_{big_rows}"
    measured[key] = {"rows": big_rows, "GB": big_rows * D * 2 / 1e9, "fill_seconds": t_fill,
                     "layout": f"{big_rows // block_rows} copies of the {block_rows}-row generated block (larger than any last-level cache: the scan streams from DRAM)",
                     "fair": fair_on(base), "reference_faithful": faithful_on(base)}
    big = measured[key]
    del base
    model = ""
    try:
        model = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
    except Exception:  # noqa: BLE001
        pass
    scale = big_rows / n_rows
    return {
        "value": big["fair"]["queries_per_s"] * scale,
        "unit": f"queries/s over {n_rows:.0e} rows (fair rate measured at {big_rows:.0e} DRAM rows x {scale:g})",
        "unit_note": "the scan is linear in rows and both sizes stream from DRAM; `measured` holds the unscaled figures",
        "cores": cores,
        "cores_visible": visible,
        "cpu_time_quota_cores": quota,
        "cores_note": "threads used = 2 x the cgroup's CPU-time quota (cpu.max): under a quota the group is throttled per period, and "
                      "oversubscribing by 2 measured best (8 / 32 / 64 / 256 threads -> 77 / 100 / 58 / 20 queries/s on a 1e6-row sample, round 3)",
        "kind": "port",
        "cpu_model": model,
        "sample": f">= {min_seconds:g} s per mode; N={big_rows:.0e} in DRAM: fair {big['fair']['queries']} queries on {cores} threads in "
                  f"{big['fair']['seconds']:.1f} s, reference-faithful {big['reference_faithful']['queries']} in {big['reference_faithful']['seconds']:.1f} s; "
                  f"N=1e5: 1000 distinct queries, 3 modes",
        "scan_GBps": big["fair"]["scan_GBps"],
        "measured": measured,
        "reference_faithful": {"value": big["reference_faithful"]["queries_per_s"] * scale, "unit": f"queries/s over {n_rows} rows, measured at {big_rows} rows x {scale:g}",
                               "cores": 1, "scan_GBps": big["reference_faithful"]["scan_GBps"]},
    }


def concurrent_callers_bench(vecs, k, headline_qps, thread_counts=(64, 512), rounds=12):
    """The reference's call shape at the metric's size: T host threads, ONE query per call, host pointers in and out
    (src/main.rs:896-934,1043-1049; src/query_disk_index.rs:711-736), through the cross-thread coalescer of the C ABI
    (mse_dispatcher, csrc/dispatch.hip).  The callers are native threads (scripts/native/mse_callers.c -- what a Rust host's
    thread-per-core loop looks like to the library), closed loop: each issues its next query when the previous one is answered.
    Every answer is checked against one batched device pass over the same queries (a separate searcher, matrix-core mode)."""
    import ctypes as C
    import subprocess
    import numpy as np
    import mse
    from mse import ffi
    so = os.path.join(ROOT, "scripts", "native", "libmse_callers.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.dirname(so)], stdout=subprocess.DEVNULL)
    H = C.CDLL(so)
    H.mse_callers_run.restype = C.c_double
    H.mse_callers_run.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t,
                                  C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_int)]
    fn = C.cast(ffi.lib().mse_dispatcher_topk_f16, C.c_void_p)
    n_max = max(thread_counts) * rounds
    qdev = mse.VectorList.generate(SEED_QUERY, 1 << 20, n_max, D)       # fresh queries, copied to the HOST: callers hand over host pointers
    q = qdev.rows(0, n_max)
    checker = mse.Searcher(vecs)
    want_s, want_i = checker.bruteforce_topk(q, k, mse.MODE_MFMA)
    checker.close()
    qdev.close()
    disp = mse.Dispatcher(vecs)                                          # defaults: one full pass (320 queries) per gather, wait budget from the row count
    disp.search(q[0], k)                                                 # lone caller: answered at once; allocates the worker's scratch
    points = []
    for T in thread_counts:
        n = T * rounds
        sc = np.empty((n, k), np.int64)
        ids = np.empty((n, k), np.uint32)
        lat = np.zeros(n, np.float64)
        failed = C.c_int(0)
        H.mse_callers_run(fn, disp._h, q.ctypes.data, min(n, 2 * T), D * 2, k, T, sc.ctypes.data, 8, ids.ctypes.data, 4, lat.ctypes.data, C.byref(failed))  # warm: two rounds
        st0 = disp.stats()
        dt = H.mse_callers_run(fn, disp._h, q.ctypes.data, n, D * 2, k, T, sc.ctypes.data, 8, ids.ctypes.data, 4, lat.ctypes.data, C.byref(failed))
        st1 = disp.stats()
        ok = bool(dt > 0 and failed.value == 0 and np.array_equal(ids, want_i[:n]) and np.array_equal(sc, want_s[:n]))
        passes = st1["passes"] - st0["passes"]
        points.append({"threads": T, "queries": n, "queries_per_s": n / dt if dt > 0 else None, "seconds": dt,
                       "latency_ms": {"p50": float(np.percentile(lat, 50)), "p99": float(np.percentile(lat, 99)), "max": float(lat.max())},
                       "passes": passes, "queries_per_pass": n / max(passes, 1), "all_answers_equal_batched_pass": ok,
                       "vs_resident_batch_headline": (n / dt / headline_qps) if dt > 0 and headline_qps else None})
    st = disp.stats()
    disp.close()
    best = max(points, key=lambda p: p["queries_per_s"] or 0)
    return {"metric": "queries/s through host pointers, one query per call from T native threads (closed loop), coalesced behind the C ABI",
            "threads": best["threads"], "queries_per_s": best["queries_per_s"], "latency_ms": best["latency_ms"],
            "vs_resident_batch_headline": best["vs_resident_batch_headline"], "points": points, "k": k,
            "dispatcher": {"max_queries_per_pass": int(ffi.lib().mse_queries_per_pass_max(D)), "passes_started_by_wait_budget": st["deadline_fires"], "requests_repeated_alone": st["retried_alone"]},
            "config": {"workload": f"{len(vecs)} x {D} fp16 rows resident; each call: 1 query of {D} f16 from host memory in, top-{k} (i64 scores, u32 ids) to host memory out"}}


def index_callers_bench(k, threads=64, rounds=40, n=100_000):
    """BASELINE configs[0]'s index -- 1e5 x 1152, top-10 through the in-memory index surface (src/main.rs:815-934: fp32 rows added in
    1024-row batches, SQfp16-IP search) -- in the reference's call shape: `threads` native threads, ONE fp32 query per
    mse_index_search call (the per-request `index.search(&query, k)` under the shared read guard, :900,:1046), closed loop.  The calls
    coalesce inside the index; every answer (labels and float distances) is checked against one batched search of the same queries."""
    import ctypes as C
    import subprocess
    import numpy as np
    import mse
    from mse import ffi
    so = os.path.join(ROOT, "scripts", "native", "libmse_callers.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.dirname(so)], stdout=subprocess.DEVNULL)
    H = C.CDLL(so)
    H.mse_callers_run.restype = C.c_double
    H.mse_callers_run.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t,
                                  C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_int)]
    rng = np.random.default_rng(40)
    x = (rng.standard_normal((n, D)) / np.sqrt(D)).astype(np.float32)
    idx = mse.ScalarQuantizerIndex(D)
    for lo in range(0, n, 1024):                      # INDEX_ADD_BATCH (src/main.rs:815)
        idx.add(x[lo:lo + 1024])
    nq = threads * rounds
    q = rng.standard_normal((nq, D)).astype(np.float32)
    want = idx.search(q, k)
    one = idx.search(q[:1], k)                        # a lone caller
    fn = C.cast(ffi.lib().mse_index_search, C.c_void_p)
    dist = np.empty((nq, k), np.float32)
    lab = np.empty((nq, k), np.int64)
    lat = np.zeros(nq, np.float64)
    failed = C.c_int(0)
    H.mse_callers_run(fn, idx._h, q.ctypes.data, min(nq, 4 * threads), D * 4, k, threads, dist.ctypes.data, 4, lab.ctypes.data, 8, lat.ctypes.data, C.byref(failed))
    st0 = idx.stats()
    dt = H.mse_callers_run(fn, idx._h, q.ctypes.data, nq, D * 4, k, threads, dist.ctypes.data, 4, lab.ctypes.data, 8, lat.ctypes.data, C.byref(failed))
    st1 = idx.stats()
    t0 = time.perf_counter()
    for i in range(64):
        idx.search(q[i:i + 1], k)
    t_alone = (time.perf_counter() - t0) / 64
    ok = bool(dt > 0 and failed.value == 0 and np.array_equal(lab, want.labels) and np.array_equal(dist, want.distances) and
              np.array_equal(one.labels[0], want.labels[0]))
    passes = st1["passes"] - st0["passes"]
    idx.close()
    return {"metric": "configs[0] index (1e5 x 1152 fp32 rows -> SQfp16, top-10) searched by T native threads x 1 fp32 query per mse_index_search call",
            "threads": threads, "queries": nq, "queries_per_s": nq / dt if dt > 0 else None,
            "latency_ms": {"p50": float(np.percentile(lat, 50)), "p99": float(np.percentile(lat, 99))},
            "queries_per_pass": nq / max(passes, 1), "one_caller_ms_per_query": t_alone * 1e3, "one_caller_queries_per_s": 1.0 / t_alone,
            "all_answers_equal_batched_search": ok}


def shard_point_bench(k, nq, full_ms_per_step, steps=20, rows=12_500_000):
    """What ONE of eight GPUs runs per step when the 1e8-row index is sharded 8 ways (BASELINE configs[3]): the 12.5 M-row
    (28.8 GB) local search -- scan + tournament + exact re-score + certificate -- one ncclAllGather of its packed [Q, k] block
    (RCCL through the C ABI; a world of ONE here: no multi-GPU node was available to the build) and the k-way merge, timed on
    this GPU through the same mse_shard_group path `bench.py --gpus 8` drives.  `projected_8gpu` = T(1e8 on one GPU) / (8 x
    T(this step)): what the strong-scaling efficiency would be if the exchange cost what it costs in a world of one (it is
    30 KB per rank per step; the xGMI leg is unmeasured)."""
    import torch
    import mse
    group = mse.ShardGroup(1, D, devices=[0])
    group.generate(SEED_BASE, 0, rows)
    label = "ONE ncclAllGather per step through librccl (world of one)"
    try:
        with _stdout_to_stderr():
            group.set_exchange(group.EXCHANGE_RCCL)
    except mse.MseError as e:
        label = f"peer-store exchange (RCCL unavailable: {e})"
    qs = mse.VectorList.generate(SEED_QUERY, 1 << 21, nq * 4, D)
    out_s = torch.empty((nq, k), dtype=torch.int64, device="cuda")
    out_i = torch.empty((nq, k), dtype=torch.int32, device="cuda")
    searcher = group.searcher(0)

    def step(i):
        group.bruteforce_topk_dev(qs.device_ptr + (i % 4) * nq * D * 2, nq, k, out_s.data_ptr(), out_i.data_ptr(), mse.MODE_MFMA)

    with _stdout_to_stderr():
        for i in range(3):
            step(i)
    searcher.scan_timing(2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps * 1e3
    scan_ms, n_scan = searcher.scan_timing(0)
    t = group.last_timing()
    scan = scan_ms / max(n_scan, 1)
    out = {"rows_per_gpu": rows, "queries_per_step": nq, "ms_per_step": dt, "steps": steps, "exchange": label,
           "step_breakdown": {"scan_ms": scan, "tail_ms": t["local_search_ms"] - scan, "exchange_ms": t["exchange_ms"], "merge_ms": t["merge_ms"],
                              "wall_ms": t["wall_ms"], "of": "last timed step (scan: HIP-event average of the timed steps)"},
           "roofline": {"bound": "hbm", "achieved": rows * D * 2 / (scan * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": rows * D * 2 / (scan * 1e-3) / 1e9 / HBM_PEAK_GBS},
           "queries_per_s_x8_if_linear": 8 * nq / (dt * 1e-3) / 8,
           "projected_8gpu": {"queries_per_s": nq / (dt * 1e-3), "efficiency_vs_one_gpu_1e8": full_ms_per_step / (8 * dt),
                              "note": "per-step time of a 12.5 M-row shard against an eighth of the one-GPU 1e8-row step; the exchange is a world-of-one all-gather"}}
    del out["queries_per_s_x8_if_linear"]
    group.close()
    qs.close()
    return out


def pq_bench(args):
    """BASELINE configs[4] shape at BASELINE.md's size: full ADC scan of 1e8 x 64-byte OPQ codes (+4 descriptor bytes), top-200 by
    approximate score (the re-rank candidates), all arrays resident in HBM.  End to end per query (query upload, table build, scan
    keeping one maximum per 64 vectors, tournament, re-score of the best groups, exact top-r, download): one query per call, and
    32 queries per call (one upload / download; the queries go through in EIGHTS that share one pass over the codes --
    pq_scan64x4_kernel<.., 8>, 8-bit integer nomination on the matrix cores under a certificate -- and the groups alternate between
    two streams).  Both windows are
    at least a second long.  `roofline` is the scan KERNEL's (68 B per vector per launch / its HIP-event duration); the end-to-end
    figure (a pass taken as four batched per-query times) stands beside it."""
    import numpy as np
    import mse
    n = int(args.pq_rows)
    rng = np.random.default_rng(0)
    cents = (rng.standard_normal((256, D)) / np.sqrt(D)).astype(np.float32)
    T = np.linalg.qr(rng.standard_normal((D, D)))[0].astype(np.float32)
    pq = mse.ProductQuantizer(cents, T, 18, D)
    # 6.4 GB of random code rows in seconds: distinct byte-mask copies of one random block
    blk = min(n, 1_000_000)
    block = rng.integers(0, 256, size=(blk, 64), dtype=np.uint8)
    codes = np.empty((n, 64), np.uint8)
    for c0 in range(0, n, blk):
        m = min(blk, n - c0)
        np.bitwise_xor(block[:m], rng.integers(0, 256, size=64, dtype=np.uint8), out=codes[c0:c0 + m])
    desc = np.resize(rng.integers(0, 256, size=(blk, 4), dtype=np.uint8), (n, 4))
    gc = mse.Codes(codes, desc)
    del codes, desc, block
    scales = np.array([0.5, 0, -0.25, 0], np.float32) / np.float32(512)
    qs = (rng.standard_normal((32, D)) / np.sqrt(D)).astype(np.float32)
    pq.scan_topk(gc, qs[0], 200, 10, None, scales)
    t0, calls1 = time.perf_counter(), 0
    while calls1 < 40 or time.perf_counter() - t0 < 1.0:
        pq.scan_topk(gc, qs[calls1 % 32], 200, 10, None, scales)
        calls1 += 1
    dt = (time.perf_counter() - t0) / calls1
    for _ in range(2):
        pq.scan_topk_batch(gc, qs, 200, 10, None, scales)
    t0, calls = time.perf_counter(), 0
    while calls < 40 or time.perf_counter() - t0 < 1.0:
        pq.scan_topk_batch(gc, qs, 200, 10, None, scales)
        calls += 1
    wall = time.perf_counter() - t0
    db = wall / (calls * len(qs))
    uncert = pq.last_uncertified
    # the same at 64 queries per call (what the cross-thread coalescer hands over at most): start-up and drain of a call weigh half
    qs64 = np.concatenate([qs, qs[::-1]])
    pq.scan_topk_batch(gc, qs64, 200, 10, None, scales)
    t0, calls64 = time.perf_counter(), 0
    while calls64 < 20 or time.perf_counter() - t0 < 0.6:
        pq.scan_topk_batch(gc, qs64, 200, 10, None, scales)
        calls64 += 1
    db64 = (time.perf_counter() - t0) / (calls64 * 64)
    # the scan kernel by itself: calls of EIGHT queries are one group on one stream (nothing beside the scan), HIP events per launch
    pq.scan_timing(2)
    for i in range(24):
        pq.scan_topk_batch(gc, qs[8 * (i % 4):8 * (i % 4) + 8], 200, 10, None, scales)
    k_ms, k_n = pq.scan_timing(0)
    # ... and SUSTAINED: 64-query calls = eight scans back to back, alternating between two streams with the tails beside them;
    # first scan's start to last scan's end over the scans of each call (mse_pq_scan_sustained)
    pq.scan_timing(2)
    for _ in range(12):
        pq.scan_topk_batch(gc, qs64, 200, 10, None, scales)
    span_ms, span_n = pq.scan_sustained()
    pq.scan_timing(0)
    sus_ms = span_ms / max(span_n, 1)
    gbs_sustained = n * 68 / (sus_ms * 1e-3) / 1e9 if span_n else None
    per_pass = 8
    gbs_pass = n * 68 / (per_pass * db) / 1e9
    k_avg = k_ms / max(k_n, 1)
    gbs_kernel = n * 68 / (k_avg * 1e-3) / 1e9 if k_n else None
    traffic = None
    try:     # HBM bytes per scan launch from the PMC passes (collected offline: a counter pass cannot run inside a timed run)
        pm = json.load(open(PMC_TRAFFIC))["pq_scan64x4"]
        if pm["vectors"] == n:
            traffic = pm["hbm_read_bytes_per_launch"] + pm["hbm_write_bytes_per_launch"]
    except Exception:  # noqa: BLE001
        pass
    return {"metric": "OPQ/PQ 64x8-bit ADC scan + top-200", "ms_per_query": dt * 1e3, "ms_per_query_batched": db * 1e3,
            "queries_per_s_batched": 1.0 / db, "queries_per_call_batched": len(qs), "queries_per_pass_batched": per_pass, "vectors": n,
            "timed": {"one_query_calls": calls1, "batched_calls": calls, "batched_seconds": wall, "calls_of_64": calls64},
            "at_64_per_call": {"ms_per_query": db64 * 1e3, "queries_per_s": 1.0 / db64, "end_to_end_frac": n * 68 / (8 * db64) / 1e9 / HBM_PEAK_GBS},
            "uncertified_queries_last_batch": uncert,
            "codes_GBps_end_to_end_one_query_per_call": n * 68 / dt / 1e9, "unit": "GB/s of codes + descriptor bytes",
            "roofline": {"bound": "hbm", "kernel": "pq_scan64x4_kernel<16, 8>", "achieved": gbs_sustained, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (gbs_sustained / HBM_PEAK_GBS) if gbs_sustained else None, "avg_launch_ms": sus_ms, "launches_timed": span_n,
                         "frac_is": "SUSTAINED: scans back to back (64-query calls: eight scans on two streams, tails beside them), first scan's "
                                    "start to last scan's end / scans; profiles/r05_pq_scan_stats.txt holds the rocprofv3 trace of the same calls",
                         "burst": {"achieved": gbs_kernel, "frac": (gbs_kernel / HBM_PEAK_GBS) if gbs_kernel else None, "avg_launch_ms": k_avg, "launches_timed": k_n,
                                   "note": "one scan per call with nothing before or beside it (eight-query calls): the device has paused before every launch"},
                         "bytes_per_launch": n * 68, "queries_per_launch": per_pass, "traffic": traffic,
                         "traffic_source": ("profiles/" + os.path.basename(PMC_TRAFFIC) + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE)") if traffic else None,
                         "end_to_end": {"achieved": gbs_pass, "frac": gbs_pass / HBM_PEAK_GBS,
                                        "note": "68 B x vectors / (8 x batched per-query time): table build, scan, tournament, re-score of the nominated "
                                                "groups, exact top-r, certificate, download -- two streams, one group of eight queries each"},
                         "note": "achieved = 68 B x vectors per launch / the kernel's sustained cost per launch (above); `burst` = its HIP-event duration "
                                 "in 24 eight-query calls.  Kernel: bank-conflict-free rotated gathers, sums on the matrix cores "
                                 "(v_mfma_i32_16x16x64_i8), eight queries per pass with 8-bit tables under a certificate -- DESIGN.md 3.3"},
            "config": {"workload": f"{n} x (64 B codes + 4 B descriptors), eight queries' 8-bit tables (code-major, 138 KiB) in LDS, r = 200"}}


from bench_ann import train_codec, easy_generator as clustered_generator  # noqa: E402  (the synthetic sets live in bench_ann.py)


def graph_rows(n, seed, centres):
    import numpy as np
    g = np.random.default_rng(seed)
    asg = g.integers(0, len(centres), n)
    x = centres[asg] + g.standard_normal((n, D)).astype(np.float32) * np.float32(0.3 / np.sqrt(D))
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return x.astype(np.float16)


def graph_centres(n):
    import numpy as np
    c = np.random.default_rng(0).standard_normal((max(64, n // 50), D)).astype(np.float32)
    return c / np.linalg.norm(c, axis=1, keepdims=True)


def graph_bench(args):
    """The graph side of the index.  Build: the Vamana passes of generate_index_shard
    (diskann/src/lib.rs:287-324: random fill, one pass at the default relaxation factor 65536, R = 64, L = 192, C = 750; the
    optional second pass `-s` is timed on a copy) on the device over a synthetic clustered set.  Search, on the one-pass graph
    (the tool's default): query_disk_index::greedy_search
    (src/query_disk_index.rs:144-212) GPU-resident and batched, neighbours scored exactly (the vectors are in HBM), and
    diskann::greedy_search (lib.rs:183-211) batched the same way; recall@10 against the exact brute-force top-10."""
    import numpy as np
    import mse
    n, nq, R, K = int(args.graph_rows), 1024, 64, 10
    centres = graph_centres(n)
    base = graph_rows(n, 1, centres)
    qh = graph_rows(nq, 2, centres)
    vl = mse.VectorList.from_f16s(base.view(np.uint16), D)
    searcher = mse.Searcher(vl)
    med = mse.medioid(vl)
    g = mse.BuildGraph(n, R)
    g.random_fill(1)
    order = np.random.default_rng(3).permutation(n).astype(np.uint32)
    batch = 2048
    g.build(searcher, order[:batch], med, mse.IndexBuildConfig(r=R, l=192, maxc=750), batch)   # warm-up: one batch
    t0 = time.perf_counter()
    g.build(searcher, order, med, mse.IndexBuildConfig(r=R, l=192, maxc=750), batch)
    t1 = time.perf_counter()
    host = g.to_host()                                   # the searches below run on this graph: the tool's default is ONE pass
    g2 = mse.BuildGraph(n, R, host)                      # the optional second pass (-s), timed on a copy
    t1b = time.perf_counter()
    g2.build(searcher, order, med, mse.IndexBuildConfig(r=R, l=192, maxc=750), batch)   # -B defaults to 65536 as well
    t_second = time.perf_counter() - t1b
    g2.close()
    build = {"metric": "Vamana build (diskann::build_graph), points/s", "first_pass_points_per_s": n / (t1 - t0),
             "second_pass_points_per_s": n / t_second, "batch": batch, "r": R, "l": 192, "maxc": 750,
             "mean_degree": float(host.deg.mean())}
    # OPQ-shaped 64 x 256 codec trained on a 20 000-row sample; codes by quantize_batch on the device
    rng = np.random.default_rng(4)
    samp = base[rng.choice(n, min(n, 20000), replace=False)].astype(np.float32)
    cents, T = train_codec(samp)
    pq = mse.ProductQuantizer(cents, T, 18, D)
    codes_h = np.concatenate([pq.quantize_batch(base[s0:s0 + 8192].astype(np.float32)) for s0 in range(0, n, 8192)])
    codes = mse.Codes(codes_h, None)
    dgraph = mse.DeviceGraph(host)
    starts = np.full(nq, med, np.uint32)
    _, truth = searcher.bruteforce_topk(qh.view(np.uint16), K)
    qf = qh.astype(np.float32)
    # BASELINE configs[4]: flat ADC scan of the codes, best 200 by approximate score re-scored exactly in fp16, top-10
    pq.scan_topk(codes, qf[0], 200, K, searcher)
    t0 = time.perf_counter()
    rr = [pq.scan_topk(codes, qf[i], 200, K, searcher)[1] for i in range(128)]
    drr = (time.perf_counter() - t0) / 128
    rerank = {"metric": "OPQ/PQ 64x8-bit flat scan, top-200 by ADC re-scored exactly, top-10", "ms_per_query": drr * 1e3,
              "recall_at_10": sum(len(set(rr[i].tolist()) & set(truth[i].tolist())) for i in range(128)) / (K * 128), "rows": n}
    out = []
    for L in (64, 200):
        # timed: the C-ABI call with host arrays in and out (one warm call first: scratch is allocated on first use)
        mse.disk_search_batch(searcher, pq, codes, dgraph, starts, qh.view(np.uint16), None, None, True, 4, L, 1024, as_arrays=True)
        t0 = time.perf_counter()
        res = mse.disk_search_batch(searcher, pq, codes, dgraph, starts, qh.view(np.uint16), None, None, True, 4, L, 1024, as_arrays=True)
        dt = time.perf_counter() - t0
        top = mse.topk_of_visited(res, K)
        hits = sum(len(set(top[i].tolist()) & set(truth[i].tolist())) for i in range(nq))
        g.search_batch(searcher, med, qh.view(np.uint16), L, as_arrays=True)
        t0 = time.perf_counter()
        rid, _, _, _ = g.search_batch(searcher, med, qh.view(np.uint16), L, as_arrays=True)
        dr = time.perf_counter() - t0
        rhits = sum(len(set(rid[i, :K].tolist()) & set(truth[i].tolist())) for i in range(nq))
        # the production setting: neighbours scored by ADC from the codes (f32 queries in, tables made on the device)
        mse.disk_search_batch(searcher, pq, codes, dgraph, starts, qf, None, None, False, 4, L, 1024, as_arrays=True)
        t0 = time.perf_counter()
        ra = mse.disk_search_batch(searcher, pq, codes, dgraph, starts, qf, None, None, False, 4, L, 1024, as_arrays=True)
        da = time.perf_counter() - t0
        atop = mse.topk_of_visited(ra, K)
        ahits = sum(len(set(atop[i].tolist()) & set(truth[i].tolist())) for i in range(nq))
        out.append({"search_list": L, "beamwidth": 4, "queries_per_s": nq / dt, "recall_at_10": hits / (K * nq),
                    "node_fetches_per_query": float(res["cmps"].mean()),
                    "adc_queries_per_s": nq / da, "adc_recall_at_10": ahits / (K * nq),
                    "in_ram_greedy_search_queries_per_s": nq / dr, "in_ram_recall_at_10": rhits / (K * nq)})
    return {"metric": "GPU-resident beam search (query_disk_index::greedy_search), batch of 1024 queries",
            "config": {"workload": f"{n} x {D} fp16 clustered rows, Vamana graph built on the device (R 64, L 192, one pass: the default of generate-index-shard), "
                                   "queries_per_s: exact neighbour scoring (disable_pq: the vectors are in HBM); adc_*: neighbours scored from the "
                                   "64-byte codes of a codec trained on a 20k sample; host arrays in / out"},
            "build": build, "pq_rerank": rerank, "results": out}


def pq_rerank_leg(rows, vecs, s, queries, truth, K=10, r=200, per_call=32):
    """BASELINE configs[4] as specified, on a quantisable set resident in HBM: OPQ/PQ 64 x 8-bit codes of the rows (codec trained on
    a 20 000-row sample, codes made on the device), flat ADC scan of ALL codes, the r best by approximate score re-scored exactly
    in fp16, top-K; recall@K against `truth` (the exact brute-force answers of the same run)."""
    import numpy as np
    import torch
    import mse
    n, nq = len(vecs), queries.shape[0]
    rng = np.random.default_rng(4)
    sel = torch.from_numpy(np.sort(rng.choice(n, min(n, 20000), replace=False))).cuda()
    cents, T = train_codec(rows[sel].float().cpu().numpy())
    pq = mse.ProductQuantizer(cents, T, 18, D)
    t0 = time.perf_counter()
    codes = mse.Codes.quantize_base(pq, vecs)
    t_quant = time.perf_counter() - t0
    qf = queries.float().cpu().numpy()
    pq.scan_topk_batch(codes, qf[:per_call], r, K, s)
    t0 = time.perf_counter()
    got = np.concatenate([pq.scan_topk_batch(codes, qf[i:i + per_call], r, K, s)[1] for i in range(0, nq, per_call)])
    dt = time.perf_counter() - t0
    rec = sum(len(set(got[i].tolist()) & set(truth[i].tolist())) for i in range(nq)) / (K * nq)
    out = {"metric": "OPQ/PQ 64x8-bit flat scan of all codes, top-%d by ADC re-scored exactly (fp16 rows), top-%d" % (r, K), "rows": n,
           "queries": nq, "queries_per_call": per_call, "queries_per_s": nq / dt, "ms_per_query": dt / nq * 1e3, "recall_at_10": rec, "r": r,
           "uncertified_queries_last_batch": pq.last_uncertified,
           "codes": {"made_on_device_seconds": t_quant, "vectors_per_s": n / t_quant, "codec": "64 x 256, rotation + per-subspace k-means on a 20 000-row sample (3 iterations)"}}
    codes.close()
    return out


def ann_scale_bench(args):
    """A driver-timed approximate-search figure AT THE METRIC'S SIZE: 1e8 x 1152 clustered rows resident in HBM (230 GB), searched by
    the OPQ/PQ flat scan + fp16 exact re-rank pipeline (BASELINE configs[4]); recall@10 against the exact brute-force answers over
    the same rows, 1024 queries.  (The graph index at this size needs a 10-20 minute build: profiles/r04_graph_index_1e8*.json.)"""
    import numpy as np
    import torch
    import mse
    from mse import ffi
    n, nq, K = int(args.ann_rows), 1024, 10
    free_b, total_b = ffi.sz(), ffi.sz()
    ffi.check(ffi.lib().mse_device_mem_info(free_b, total_b))
    need = n * D * 2 + n * 68 + (16 << 30)
    if need > free_b.value:
        return {"skipped": f"{n} rows need {need / 1e9:.0f} GB, {free_b.value / 1e9:.0f} GB free"}
    clustered = clustered_generator(n)
    t0 = time.perf_counter()
    rows, queries = clustered(n, 1), clustered(nq, 2)
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t0
    vecs = mse.VectorList.wrap_device(rows.data_ptr(), n, D, keepalive=rows)
    s = mse.Searcher(vecs)
    qh = queries.cpu().numpy().view(np.uint16)
    s.bruteforce_topk(qh[:8], K)
    t0 = time.perf_counter()
    _, truth = s.bruteforce_topk(qh, K)
    t_exact = time.perf_counter() - t0
    out = pq_rerank_leg(rows, vecs, s, queries, truth, K)
    out["metric"] = f"queries/sec over a {n:.0e}x1152 index @ recall@10>=0.95: OPQ/PQ flat scan + fp16 exact re-rank (recall against the exact answers of the same run)"
    out["value"] = out["queries_per_s"]
    out["unit"] = "queries/s"
    out["exact_brute_force_same_rows_queries_per_s"] = nq / t_exact
    out["config"] = {"workload": f"{n} x {D} fp16 hierarchical synthetic clusters generated on the device in {t_gen:.1f} s; host arrays in and out"}
    s.close()
    vecs.close()
    return out


def cpu_graph_build(n, points=192):
    """CPU side of the build line: the oracle's build_graph (batch form) on a bounded sample of the same workload --
    `points` insertions into the same random initial graph over the same rows, one thread."""
    import numpy as np
    from oracle import orc
    centres = graph_centres(n)
    base = graph_rows(n, 1, centres).view(np.uint16)
    adj, deg = orc.random_fill_graph(1, n, 64)
    order = np.random.default_rng(3).permutation(n).astype(np.uint32)[:points]
    med = int(orc.medioid(base[:20000]))   # any fixed start node serves the timing; the full medioid is a scan of its own
    t0 = time.perf_counter()
    orc.build_graph(base, adj, deg, order, med, orc.BuildConfig.make(r=64, l=192, maxc=750), points)
    dt = time.perf_counter() - t0
    return {"value": points / dt, "unit": "points/s", "cores": 1, "kind": "port",
            "sample": f"{points} insertions (one batch) into the random initial graph over the same {n} rows"}


def siglip_bench(args, world, rank, dist=None):
    """BASELINE configs[1]: SigLIP-SO400M/14-384 image tower, batch 256 random 384x384 images, bf16, one
    replica per GPU.  Random-init weights of the named architecture (no checkpoint offline); images already
    resident in HBM as fp16 NCHW (what clip_server's preprocessing thread hands to the model).  One step = one
    full forward of the batch incl. L2 normalisation and fp16 output rows."""
    import torch
    from mse import siglip
    cfg = dict(siglip.SO400M_384)
    batch = args.siglip_batch
    eng = siglip.SiglipImageEngine.from_state_dict(siglip.synthetic_state_dict(cfg), cfg, max_batch=batch)
    img = torch.empty((batch, 3, cfg["img_size"], cfg["img_size"]), dtype=torch.float16, device="cuda").uniform_(-1, 1)
    torch.cuda.synchronize()
    eng.encode_image_device(img.data_ptr(), batch)            # warm-up (encode_image synchronises its stream)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.siglip_steps):
        eng.encode_image_device(img.data_ptr(), batch)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)       # the control-plane group is gloo (CPU tensors)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # small batches: the query path sends ONE text (src/query_disk_index.rs:345-381) and a handful of images at a time; median of 15 calls
    import numpy as np

    def med_ms(fn, reps=15):
        fn()
        ts = []
        for _ in range(reps):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t1) * 1e3)
        return float(np.median(ts))
    latency = {"image": {str(b): med_ms(lambda b=b: eng.encode_image_device(img.data_ptr(), b)) for b in (1, 8, 32) if b <= batch}}
    server = None
    if rank == 0 and world == 1:
        try:
            server = server_bench(eng, cfg, batch)
        except Exception as e:  # noqa: BLE001
            server = {"error": repr(e)}
    eng.close()
    # text tower (clip_server.py:98): batch of 256 token rows, random-init weights; small next to the image tower
    import numpy as np
    tcfg = dict(siglip.SO400M_TEXT)
    teng = siglip.SiglipTextEngine.from_state_dict(siglip.synthetic_text_state_dict(tcfg), tcfg, max_batch=256)
    tok = np.random.default_rng(7).integers(2, tcfg["vocab_size"], size=(256, tcfg["context_length"]), dtype=np.int64)
    for _ in range(3):
        teng.encode_text(tok)
    tt0 = time.perf_counter()
    for _ in range(20):
        teng.encode_text(tok)
    text_dt = (time.perf_counter() - tt0) / 20
    latency["text"] = {str(b): med_ms(lambda b=b: teng.encode_text(tok[:b])) for b in (1, 8, 32)}
    # what a forward cannot go below at small batch: every weight read once from HBM (bf16): 27 blocks x (4 d^2 + 2 d mlp) + patch / token embedding rows used
    w_img = (27 * (4 * 1152 * 1152 + 2 * 1152 * 4304) + 588 * 1152 + 4 * 1152 * 1152 + 2 * 1152 * 4304) * 2
    w_txt = (27 * (4 * 1152 * 1152 + 2 * 1152 * 4304) + 1152 * 1152) * 2
    latency["weight_stream_floor_ms"] = {"image": w_img / 8e12 * 1e3, "text": w_txt / 8e12 * 1e3,
                                         "note": "bf16 weights of one forward read once at 8 TB/s"}
    latency["image_b1_over_floor"] = latency["image"]["1"] / latency["weight_stream_floor_ms"]["image"] if "1" in latency["image"] else None
    latency["text_b1_over_floor"] = latency["text"]["1"] / latency["weight_stream_floor_ms"]["text"]
    teng.close()
    text = {"metric": "SigLIP text-embeds/sec/GPU", "value": 256 / text_dt, "unit": "texts/s/GPU", "ms_per_batch": text_dt * 1e3,
            "config": {"workload": "SigLIP-SO400M text tower, batch 256 x 64 tokens (host token ids in, host features out)"},
            "tflops": 256 / text_dt * 27 * 1.968e9 / 1e12,
            "roofline": {"bound": "mfma", "achieved": 256 / text_dt * 27 * 1.968e9 / 1e12, "peak": 2500.0, "unit": "TFLOP/s",
                         "frac": 256 / text_dt * 27 * 1.968e9 / 1e12 / 2500.0, "flop_per_text": 27 * 1.968e9}}
    per_gpu = batch * args.siglip_steps / dt
    gflop_img = 0.988 + 27 * 24.647 + 3.9                     # 670.4 GFLOP per image
    tflops = per_gpu * gflop_img / 1e3
    # HBM bytes of one forward from the PMC passes (collected offline by scripts/profile_r04.sh: the counters cannot be read inside a timed
    # run); only reported when this run is the profiled configuration
    sig_traffic = None
    try:
        pm = json.load(open(PMC_TRAFFIC)).get("siglip")
        if pm and pm["batch"] == batch and pm["depth"] == 27:
            sig_traffic = pm["hbm_read_bytes_per_forward"] + pm["hbm_write_bytes_per_forward"]
    except Exception:  # noqa: BLE001
        sig_traffic = None
    return {"metric": "SigLIP img-embeds/sec/GPU", "value": per_gpu, "unit": "images/s/GPU", "total_images_per_s": per_gpu * world,
            "ms_per_batch": dt / args.siglip_steps * 1e3, "dtype": "bf16 (fp32 accumulate; fp16 residual stream, fp32 LayerNorm/softmax/GELU)",
            "config": {"workload": f"SigLIP-SO400M/14-384 image tower, batch {batch} random 384x384, 1 replica per GPU",
                       "weights": "random-init (seeded), architecture of ViT-SO400M-14-SigLIP-384"},
            "steps": args.siglip_steps, "scaling": "weak (replicas)", "text_tower": text, "latency_ms": latency,
            "server_images_per_s": (server or {}).get("value"), "server": server,
            "roofline": {"bound": "mfma", "achieved": tflops, "peak": 2500.0, "unit": "TFLOP/s", "frac": tflops / 2500.0,
                         "flop_per_image": gflop_img * 1e9, "traffic": sig_traffic,
                         "traffic_source": ("profiles/" + os.path.basename(PMC_TRAFFIC) + ": HBM bytes of ONE forward of this batch, all kernels (rocprofv3 --pmc FETCH_SIZE x 2 KiB, WRITE_SIZE x 1 KiB)") if sig_traffic else None,
                         "note": "the tower runs at the board's 1400 W power limit with the engine clock held at 1.83 of 2.4 GHz "
                                 "(profiles/r04_siglip_notes.txt); the library GEMM alone runs these shapes at 0.38-0.52 of the same peak "
                                 "(profiles/r02_gemm_calibration.txt); with every GEMM epilogue removed the forward is 17 % shorter -- the "
                                 "ceiling of epilogue hiding; DESIGN.md 3.4"}}


_SERVER_CLIENT = r"""
import asyncio, sys, time
import aiohttp, msgpack, numpy as np
port, w, h, per_req, n_req, in_flight, emb = (int(x) for x in sys.argv[1:8])
rng = np.random.default_rng(11)
hdr = (b"BM" + (54 + w * h * 3).to_bytes(4, "little") + bytes(4) + (54).to_bytes(4, "little") + (40).to_bytes(4, "little") +
       w.to_bytes(4, "little") + h.to_bytes(4, "little") + (1).to_bytes(2, "little") + (24).to_bytes(2, "little") + bytes(24))
images = [hdr + rng.integers(0, 256, size=w * h * 3, dtype=np.uint8).tobytes() for _ in range(per_req)]
body = msgpack.dumps({"images": images})       # what common.rs:61-66,91 sends: a msgpack map {"images": [bin, ...]}
async def main():
    async with aiohttp.ClientSession() as sess:
        async def one():
            async with sess.post(f"http://127.0.0.1:{port}/", data=body) as r:
                rows = msgpack.loads(await r.read())
                assert r.status == 200 and len(rows) == per_req and len(rows[0]) == 2 * emb, (r.status, rows if r.status != 200 else "")
        await one()
        sem = asyncio.Semaphore(in_flight)
        async def limited():
            async with sem:
                await one()
        t1 = time.perf_counter()
        await asyncio.gather(*[limited() for _ in range(n_req)])
        print("ELAPSED", time.perf_counter() - t1, len(body))
asyncio.run(main())
"""


def server_bench(eng, cfg, engine_batch):
    """The serving path the device-side BMP decode exists for (clip_server.py:131-170, src/common.rs:31-54): requests of 128
    Rust-style 384 x 384 24-bit BMPs (56.6 MB of msgpack, under the reference's 64 MiB body limit; 256 BMPs would be 113 MB and the
    reference would answer 413) POSTed over loopback TCP by a CLIENT PROCESS (four requests in flight) to the aiohttp app of
    mse.clip_server.ClipServer in this process; the answer is the msgpack array of 2304-byte fp16 rows.  End to end: HTTP + msgpack
    decode, header checks, H2D of the raw files, device BGR->RGB / flip / normalise, the tower, fp16 rows, msgpack encode.  Beside
    it: what the reference's preprocessing thread does with the same files on one host thread (PIL decode + ToTensor/Normalize/
    .half()), the stage the device decode removes."""
    import asyncio
    import subprocess
    import threading
    import numpy as np
    from aiohttp import web
    from mse.clip_server import ClipServer, preprocess_image
    w = h = cfg["img_size"]
    per_req, n_req, in_flight = 128, int(os.environ.get("MSE_BENCH_SERVER_REQS", "32")), int(os.environ.get("MSE_BENCH_SERVER_INFLIGHT", "6"))
    rng = np.random.default_rng(11)
    hdr = (b"BM" + (54 + w * h * 3).to_bytes(4, "little") + bytes(4) + (54).to_bytes(4, "little") + (40).to_bytes(4, "little") +
           w.to_bytes(4, "little") + h.to_bytes(4, "little") + (1).to_bytes(2, "little") + (24).to_bytes(2, "little") + bytes(24))
    sample = [hdr + rng.integers(0, 256, size=w * h * 3, dtype=np.uint8).tobytes() for _ in range(16)]
    t0 = time.perf_counter()
    for im in sample:
        preprocess_image(im, (w, h))
    host_decode = len(sample) / (time.perf_counter() - t0)
    eng.image_size = (w, h)
    # a second replica of the tower (856 MB of weights): two model threads, so one batch's host side (hand-off, upload of 113 MB of
    # raw files, download) runs beside the other's device side
    from mse import siglip
    n_rep = int(os.environ.get("MSE_BENCH_SERVER_REPLICAS", "2"))
    extra = []
    for _ in range(max(0, n_rep - 1)):
        e2 = siglip.SiglipImageEngine.from_state_dict(siglip.synthetic_state_dict(cfg), cfg, max_batch=engine_batch)
        e2.image_size = (w, h)
        extra.append(e2)
    srv = ClipServer({"device": "cuda:0", "model": "ViT-SO400M-14-SigLIP-384", "model_name": "siglip-so400m-14-384",
                      "max_batch_size": per_req, "port": 0}, [eng] + extra)
    srv.start_threads()
    loop = asyncio.new_event_loop()
    ready = threading.Event()
    state = {}

    def serve():
        asyncio.set_event_loop(loop)
        runner = web.AppRunner(srv.make_app())
        loop.run_until_complete(runner.setup())
        site = web.TCPSite(runner, "127.0.0.1", 0)
        loop.run_until_complete(site.start())
        state["port"] = runner.addresses[0][1]
        state["runner"] = runner
        ready.set()
        loop.run_forever()

    th = threading.Thread(target=serve, daemon=True)
    th.start()
    ready.wait(30)
    try:
        out = subprocess.run([sys.executable, "-c", _SERVER_CLIENT, str(state["port"]), str(w), str(h), str(per_req), str(n_req),
                              str(in_flight), str(cfg["emb_dim"])], capture_output=True, text=True, timeout=600)
        line = [ln for ln in out.stdout.splitlines() if ln.startswith("ELAPSED")]
        if not line:
            raise RuntimeError("client failed: " + (out.stderr or out.stdout)[-400:])
        dt, body_len = float(line[0].split()[1]), int(line[0].split()[2])
    finally:
        asyncio.run_coroutine_threadsafe(state["runner"].cleanup(), loop).result(30)
        loop.call_soon_threadsafe(loop.stop)
        srv.stop_threads()
        for t in srv._threads:
            t.join(30)
        for e2 in extra:
            e2.close()
    return {"metric": "clip_server images/s end to end (HTTP + msgpack + device BMP decode + tower + fp16 rows)",
            "value": per_req * n_req / dt, "unit": "images/s", "images_per_request": per_req, "requests": n_req, "in_flight": in_flight,
            "request_bytes": body_len, "engine_batch_capacity": engine_batch, "engine_replicas": n_rep, "client": "separate process, loopback TCP",
            "host_preprocess_images_per_s_one_thread": host_decode,
            "note": "the model thread runs the BMP jobs waiting in its queue as one engine call (up to the engine's batch capacity); "
                    "host_preprocess = PIL decode + normalise + fp16 of the same files on one thread, the reference's preprocessing_thread "
                    "(clip_server.py:131-146), which the device decode replaces"}


def emit(full, detail_path):
    """The FULL result object goes to `detail_path` (and nowhere near stdout); stdout gets ONE compact line (bench_line.py: contract
    scalars, config, roofline with flat per-leg scalars, cpu_baseline; <= 4 KB) -- what the driver's record parses."""
    import bench_line
    full["detail_file"] = os.path.relpath(detail_path, ROOT) if detail_path else None
    if detail_path:
        try:
            os.makedirs(os.path.dirname(os.path.abspath(detail_path)), exist_ok=True)
            with open(detail_path, "w") as f:
                json.dump(full, f, indent=1)
                f.write("\n")
            out_dir = os.path.join(ROOT, "gpurun_out")       # on a gpurun box this directory travels back
            if os.path.isdir(out_dir) and os.path.dirname(os.path.abspath(detail_path)) != out_dir:
                with open(os.path.join(out_dir, os.path.basename(detail_path)), "w") as f:
                    json.dump(full, f, indent=1)
        except OSError as e:
            print(f"[bench] could not write {detail_path}: {e}", file=sys.stderr)
            full["detail_file"] = None
    sys.stdout.flush()
    print(json.dumps(bench_line.compact_line(full)), flush=True)


def dry_run(args):
    """`--dry-run`: no device, no library.  Parses the launch environment as a real run would (torchrun's WORLD_SIZE / RANK against
    --gpus, the in-process shape, --logical-shards) and prints a compact line of nulls, so that the command line the driver will use for
    N = 1, 2, 4, 8 can be checked on a CPU-only box (tests/test_abi_and_host.py)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    in_process = world == 1 and args.gpus > 1
    n_gpus = args.gpus if in_process else world
    if world > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus < 1 or args.steps < 1 or args.warmup < 0:
        raise SystemExit("--gpus >= 1, --steps >= 1, --warmup >= 0")
    n_total = int(args.rows)
    nq = args.queries if args.queries > 0 else 320
    if rank != 0:
        return
    shape = "one process, a host thread per shard" if in_process else "one process per GPU (torchrun)" if world > 1 else "one GPU"
    full = {"metric": "queries/sec over 1e8x1152 index @ recall@10>=0.95 (exact brute force: recall 1.0)", "value": None, "unit": "queries/s",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f16 in, f32 accumulate, i64 fixed-point scores", "data": "dry-run (no device, nothing measured)",
            "config": {"workload": f"brute-force top-{args.k} over {n_total} x {D} fp16 rows, {nq} queries/step, row-sharded over {n_gpus} GPU(s)",
                       "rows_total": n_total, "rows_per_gpu": (n_total + n_gpus - 1) // n_gpus, "queries_per_step": nq, "k": args.k,
                       "parallelism": f"row-shard x{n_gpus}",
                       "exchange": {"kind": shape + ("; ONE ncclAllGather of the packed 12 B/record blocks per step" if n_gpus > 1 else ""),
                                    "rccl_ranks": 0} if n_gpus > 1 else None},
            "roofline": {"bound": "hbm", "kernel": "scan_mfma_kernel<2,20>", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None},
            "sharded_ann": ({"skipped": "dry run"} if n_gpus > 1 and not args.no_sharded_ann else None), "dry_run": True}
    emit(full, None)


def rccl_probe_code(n_gpus):
    """Child-process probe of the in-process RCCL exchange: tiny shards on devices 0..n-1, bring-up, one search through the all-gather."""
    return ("import sys; sys.path.insert(0, %r); import torch, numpy as np, mse\n"
            "g = mse.ShardGroup(%d, %d, devices=list(range(%d))); g.generate(1, 0, %d)\n"
            "g.set_exchange(g.EXCHANGE_RCCL); q = np.zeros((4, %d), np.uint16); q[:, 0] = 0x3c00\n"
            "s, i = g.bruteforce_topk(q, 3, mse.MODE_EXACT); assert g.rccl_ranks == %d; print('PROBE_OK')\n"
            % (os.path.join(ROOT, "meme-search-engine_amd"), n_gpus, D, n_gpus, 4096 * n_gpus, D, n_gpus))


def main():
    global T_START
    T_START = time.perf_counter()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=float, default=1e8, help="total index rows (all ranks)")
    ap.add_argument("--queries", type=int, default=0,
                    help="queries per step (one scan pass over the rows serves up to 256); 0 = pick 128 or 256 by measured queries/s")
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--logical-shards", action="store_true",
                    help="developer dry run of the in-process --gpus N path on fewer devices (shard g on device g mod count)")
    ap.add_argument("--no-siglip", action="store_true", help="skip the SigLIP image-tower leg")
    ap.add_argument("--no-pq", action="store_true", help="skip the OPQ/PQ scan leg")
    ap.add_argument("--no-shard-point", action="store_true", help="skip the 12.5 M-row shard step (what one of 8 GPUs runs)")
    ap.add_argument("--no-callers", action="store_true", help="skip the concurrent-callers leg (T threads x 1 query through the coalescer)")
    ap.add_argument("--pq-rows", type=float, default=1e8)
    ap.add_argument("--no-graph", action="store_true", help="skip the GPU-resident beam-search leg")
    ap.add_argument("--no-request-path", action="store_true", help="skip the whole-request leg (text in -> top-k out on the device; inside the hard-set graph leg)")
    ap.add_argument("--graph-rows", type=float, default=2e5)
    ap.add_argument("--no-graph-scale", action="store_true", help="skip the 1e7-row graph-index leg (a ~1 minute build)")
    ap.add_argument("--no-ann-scale", action="store_true", help="skip the 1e8-row PQ scan + re-rank leg (recall at the metric's size)")
    ap.add_argument("--ann-rows", type=float, default=1e8)
    ap.add_argument("--graph-scale-rows", type=float, default=1e7)
    ap.add_argument("--graph-passes", type=int, default=1, help="Vamana passes of the graph-scale build (generate-index-shard -s = 2)")
    ap.add_argument("--graph-batch", type=int, default=16384, help="points inserted per batch of the graph-scale build")
    ap.add_argument("--graph-entries", type=int, default=-1,
                    help="sampled entry points of the graph-scale search (0: the medioid alone; -1: max(4096, rows / 1500))")
    ap.add_argument("--graph-kinds", default="hard,easy,ood", help="synthetic sets of the graph-index leg (bench_ann.py): easy, hard, ood")
    ap.add_argument("--graph-1e8", action="store_true",
                    help="OPT-IN: the 1e8-row graph-index leg (a 10-13 minute build, guarded by --graph-1e8-budget); not part of the default command")
    ap.add_argument("--no-graph-1e8", action="store_true", help="accepted for older command lines; the leg is opt-in now (--graph-1e8)")
    ap.add_argument("--graph-1e8-budget", type=float, default=1100.0,
                    help="seconds the predicted 1e8-row build may take (two passes if both fit, else one; beyond it the leg is skipped with that reason); "
                         "never more than what is left of --time-budget")
    ap.add_argument("--time-budget", type=float, default=590.0,
                    help="seconds the whole command aims to stay within: a side leg that would not fit what is left is skipped with that reason "
                         "(--graph-1e8 raises it to 1500 unless given)")
    ap.add_argument("--detail", default=os.path.join(ROOT, "bench_detail.json"),
                    help="where the FULL result object goes (every leg's detail); stdout carries only the compact line (bench_line.py, <= 4 KB)")
    ap.add_argument("--dry-run", action="store_true",
                    help="no device: parse the arguments and the launch environment as a real run would, print a compact line of nulls (data = 'dry-run')")
    ap.add_argument("--no-sharded-ann", action="store_true", help="--gpus N > 1: skip the sharded PQ-scan / graph-index legs")
    ap.add_argument("--ann-rows-per-gpu", type=float, default=2e6, help="--gpus N > 1: rows per GPU of the sharded approximate-search legs")
    ap.add_argument("--siglip-batch", type=int, default=256)
    ap.add_argument("--siglip-steps", type=int, default=10)
    args = ap.parse_args()
    if args.graph_1e8 and not any(a.startswith("--time-budget") for a in sys.argv[1:]):
        args.time_budget = 1500.0
    if args.dry_run:
        return dry_run(args)

    import numpy as np
    import torch
    import mse
    from mse import ffi, shard

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Two launch shapes for N > 1, both ending in the same C-ABI merge (csrc/shard_group.hip):
    #   torchrun (WORLD_SIZE = N): one process per GPU, ONE ncclAllGather of the packed records per step (mse_comm_*);
    #   bare `python bench.py --gpus N`: this one process drives all N devices, a host thread per shard, records written into
    #   the root device's buffer over peer mappings (mse_shard_group_*).
    n_dev = ffi.lib().mse_device_count()
    in_process = world == 1 and args.gpus > 1
    n_gpus = args.gpus if in_process else world
    if world > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if in_process and n_dev < args.gpus and not args.logical_shards:
        raise SystemExit(f"--gpus {args.gpus} but only {n_dev} HIP device(s) visible")
    if args.logical_shards and n_dev > 0:
        local_rank %= n_dev          # developer dry run of the torchrun shape on fewer devices (RCCL then refuses: fallback path)
    torch.cuda.set_device(local_rank)
    ffi.check(ffi.lib().mse_set_device(local_rank), "mse_set_device")
    dist = None
    if world > 1:
        import torch.distributed as dist
        # control plane only (barrier, the 128-byte RCCL id, max of the timings); the data path's collective is RCCL, called
        # from C++ on the searcher's stream
        with _stdout_to_stderr():      # gloo announces its connections on fd 1
            dist.init_process_group("gloo", rank=rank, world_size=world)
            dist.barrier()

    def sync_all():
        if in_process:
            for dv in range(min(n_gpus, n_dev)):
                torch.cuda.synchronize(dv)
        else:
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    n_total = int(args.rows)
    auto_nq = args.queries <= 0
    nq, k = (256 if auto_nq else args.queries), args.k
    tile = 320 if nq > 256 else 256 if nq > 192 else 192 if nq > 128 else 128
    lo, hi = shard.shard_range(n_total, rank, n_gpus) if not in_process else shard.shard_range(n_total, 0, n_gpus)
    free_b, total_b = ffi.sz(), ffi.sz()
    ffi.check(ffi.lib().mse_device_mem_info(free_b, total_b))
    need = (hi - lo) * D * 2 + ((hi - lo) // 32 + 1) * tile * 4 + (8 << 30)
    note = ""
    if need > free_b.value:
        if n_gpus == 1 and n_total > 10_000_000:
            note = f"1e8 rows need {need / 1e9:.0f} GB > {free_b.value / 1e9:.0f} GB free; fell back to 1e7 rows (configs[2])"
            n_total = 10_000_000
            lo, hi = 0, n_total
        else:
            raise SystemExit(f"shard does not fit: need {need / 1e9:.0f} GB, free {free_b.value / 1e9:.0f} GB")

    n_batches = 4
    nq_room = max(nq, 320) if auto_nq else nq                          # the widest pass the pick below may choose
    qsets = mse.VectorList.generate(SEED_QUERY, 0, nq_room * n_batches, D)  # query batches, resident in HBM (root device)
    out_s = torch.empty((nq_room, k), dtype=torch.int64, device="cuda")
    out_i = torch.empty((nq_room, k), dtype=torch.int32, device="cuda")
    comm = group = None
    queries_pick = None
    host_exchange = None
    exchange = None
    if in_process:
        group = mse.ShardGroup(n_gpus, D, devices=[g % n_dev for g in range(n_gpus)])
        group.generate(SEED_BASE, 0, n_total)                         # every shard's rows made on its own device
        searcher = group.searcher(0)
        peers = sum(group.peer_mapped(g) for g in range(n_gpus))
        peer_kind = ("peer stores into the root device's gather buffer" if peers == n_gpus else
                     f"staged copies (hipMemcpyPeerAsync) for {n_gpus - peers} of {n_gpus} shards, peer stores for the rest")
        exchange = {"kind": "one process, a host thread per shard; " + peer_kind, "shards": n_gpus,
                    "devices": [group.device(g) for g in range(n_gpus)], "peer_mapped_shards": peers, "rccl_ranks": 0}
        # the exchange north_star names: ONE ncclAllGather of the packed records per step among the shards' devices, each rank's
        # collective issued by its shard's host thread (csrc/shard_group.hip).  If RCCL cannot be brought up (shards sharing a
        # device, no librccl, ncclCommInitAll failing) the line is still measured over the peer-store exchange, labelled as such.
        # RCCL across several devices has never run on the build's one-GPU boxes: probe it in a CHILD process under a timeout first
        # (tiny shards, bring-up + one search), so that a bootstrap that hangs costs this run two minutes and a label, not the line
        probe_err = None
        if len(set(group.device(g) for g in range(n_gpus))) == n_gpus:
            import subprocess
            probe = rccl_probe_code(n_gpus)
            try:
                pr = subprocess.run([sys.executable, "-c", probe], capture_output=True, text=True, timeout=150)
                if "PROBE_OK" not in pr.stdout:
                    probe_err = "RCCL probe failed: " + (pr.stderr or pr.stdout).strip().splitlines()[-1][:300] if (pr.stderr or pr.stdout).strip() else "RCCL probe failed"
            except subprocess.TimeoutExpired:
                probe_err = "RCCL probe (bring-up + one all-gather in a child process) did not finish in 150 s"
        try:
            if probe_err:
                raise mse.MseError(probe_err)
            with _stdout_to_stderr():      # librccl's banner goes to fd 1; stdout carries the one JSON line only
                group.set_exchange(group.EXCHANGE_RCCL)
            exchange = {"kind": "one process, a host thread per shard; ONE ncclAllGather (librccl via the C ABI) of the packed 12 B/record "
                                "blocks per step, rank g = shard g on device g", "shards": n_gpus,
                        "devices": [group.device(g) for g in range(n_gpus)], "rccl_ranks": group.rccl_ranks,
                        "bytes_per_rank_per_step": int(ffi.lib().mse_topk_block_bytes(nq, k))}
        except mse.MseError as e:
            exchange["rccl_unavailable"] = str(e)
            print(f"[bench] RCCL exchange unavailable ({e}); measuring the peer-store exchange", file=sys.stderr)

        def step(i):
            qptr = qsets.device_ptr + (i % n_batches) * nq * D * 2
            group.bruteforce_topk_dev(qptr, nq, k, out_s.data_ptr(), out_i.data_ptr(), mse.MODE_MFMA)
    else:
        vecs = mse.VectorList.generate(SEED_BASE, lo, hi - lo, D)       # shard rows made on the device
        searcher = mse.Searcher(vecs)
        if auto_nq and world == 1:
            # queries per pass by MEASURED queries/s: the 128-query pass is HBM-bound (0.72 of the peak), the 256-query pass serves
            # twice the queries from the same stream but is bound by the power budget; whichever is faster is the
            # headline, the other is reported beside it
            pick = {}
            for cand in (128, 192, 256, 320):
                for rep_i in range(4):
                    if rep_i == 1:
                        torch.cuda.synchronize()
                        tp = time.perf_counter()
                    searcher.bruteforce_topk_dev(qsets.device_ptr, cand, k, out_s.data_ptr(), out_i.data_ptr(), mse.MODE_MFMA, id_offset=lo)
                torch.cuda.synchronize()
                pick[cand] = cand * 3 / (time.perf_counter() - tp)
            nq = max(pick, key=pick.get)
            tile = 320 if nq > 256 else 256 if nq > 192 else 192 if nq > 128 else 128
            queries_pick = {"rule": "the fastest of 128 / 192 / 256 / 320 queries per pass over 3 timed passes each", "queries_per_s": pick, "chosen": nq}
        if world > 1:
            # the exchange of the product path: RCCL through the C ABI.  Every rank reports whether its communicator came up; if any
            # did not (no usable bootstrap interface, ...) ALL ranks fall back to carrying the same packed blocks over the gloo
            # control plane and merging them with the same device kernel -- slower, loudly labelled, but the line is still measured.
            err = ""
            if os.environ.get("MASTER_ADDR", "") in ("127.0.0.1", "localhost", "::1"):
                # one node, rendezvous on loopback: let RCCL's bootstrap use the loopback interface too when nothing else is
                # named (boxes without a routable interface otherwise fail with "no socket interface found")
                os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
            with _stdout_to_stderr():
                try:
                    ids = [mse.Comm.unique_id() if rank == 0 else None]
                except Exception as e:  # noqa: BLE001
                    ids, err = [None], repr(e)
                dist.broadcast_object_list(ids, src=0)
                if ids[0] is not None:
                    try:
                        comm = mse.Comm(ids[0], rank, world)
                        if comm.size != world:
                            err = f"RCCL reports {comm.size} ranks, expected {world}"
                    except Exception as e:  # noqa: BLE001
                        err = repr(e)
                elif not err:
                    err = "rank 0 could not create the RCCL id"
            flags = [None] * world
            dist.all_gather_object(flags, err)
            if any(flags):
                if comm is not None:
                    comm.close()
                    comm = None
                host_exchange = next(f for f in flags if f)
                print(f"[bench] RCCL exchange unavailable ({host_exchange}); falling back to a gloo all-gather of the packed blocks", file=sys.stderr)
                B_blk = int(ffi.lib().mse_topk_block_bytes(nq, k))
                blk = torch.zeros(B_blk, dtype=torch.uint8, device="cuda")
                gathered_dev = torch.empty(world * B_blk, dtype=torch.uint8, device="cuda")
                gathered_host = [torch.empty(B_blk, dtype=torch.uint8) for _ in range(world)]
                exchange = {"kind": "FALLBACK: gloo all-gather of the packed 12 B/record blocks through host memory + device merge "
                                    f"(RCCL init failed: {host_exchange})", "ranks": world, "bytes_per_rank_per_step": B_blk}
            else:
                exchange = {"kind": "one process per GPU; ONE ncclAllGather (librccl via the C ABI) of the packed 12 B/record blocks per step",
                            "rccl_ranks": comm.size, "bytes_per_rank_per_step": int(ffi.lib().mse_topk_block_bytes(nq, k))}

        def step(i):
            qptr = qsets.device_ptr + (i % n_batches) * nq * D * 2
            if host_exchange is not None:
                host_step(qptr, nq, mse.MODE_MFMA, out_s, out_i)
            elif comm is None:
                searcher.bruteforce_topk_dev(qptr, nq, k, out_s.data_ptr(), out_i.data_ptr(), mse.MODE_MFMA, id_offset=lo)
            else:
                comm.search_dev(searcher, qptr, nq, k, out_s.data_ptr(), out_i.data_ptr(), mse.MODE_MFMA, id_offset=lo)

        def host_step(qptr, n_q, mode, dst_s, dst_i):
            # fallback exchange: local search into the packed block, gloo all-gather through host memory, packed merge on the device
            bb = int(ffi.lib().mse_topk_block_bytes(n_q, k))
            searcher.bruteforce_topk_dev(qptr, n_q, k, blk.data_ptr(), blk.data_ptr() + n_q * k * 8, mode, id_offset=lo)
            torch.cuda.synchronize()
            parts = [torch.empty(bb, dtype=torch.uint8) for _ in range(world)]
            dist.all_gather(parts, blk[:bb].cpu())
            gathered_dev[:world * bb].copy_(torch.cat(parts))
            ffi.check(ffi.lib().mse_merge_topk_packed_dev(searcher._h, gathered_dev.data_ptr(), world, n_q, k, dst_s.data_ptr(), dst_i.data_ptr()))

    with _stdout_to_stderr():          # anything a library prints on first use (RCCL at its first collective) stays off stdout
        for i in range(args.warmup):
            step(i)
    searcher.scan_timing(2)
    sync_all()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    sync_all()
    elapsed = time.perf_counter() - t0
    scan_ms, scan_launches = searcher.scan_timing(0)
    stats = searcher.last_stats()
    # where a step goes (last step of the timed loop): the scan kernel, the rest of the local search (pack + tournament + exact
    # re-score + certificate), the exchange leg and the merge
    breakdown = exchange_alt = None
    try:
        if in_process:
            t = group.last_timing()
            breakdown = {"scan_ms": scan_ms / max(scan_launches, 1), "local_search_ms": t["local_search_ms"],
                         "tail_ms": t["local_search_ms"] - scan_ms / max(scan_launches, 1), "exchange_ms": t["exchange_ms"],
                         "merge_ms": t["merge_ms"], "wall_ms": t["wall_ms"], "of": "last timed step; slowest shard per leg"}
            if group.exchange == group.EXCHANGE_RCCL:
                # the same index over the peer-store exchange, timed right after the headline (second variant, same line)
                group.set_exchange(group.EXCHANGE_PEER)
                for i in range(2):
                    step(i)
                sync_all()
                tb = time.perf_counter()
                for i in range(args.steps):
                    step(args.warmup + i)
                sync_all()
                dtb = time.perf_counter() - tb
                tp = group.last_timing()
                exchange_alt = {"kind": "peer stores into the root device's gather buffer (no collective)",
                                "peer_mapped_shards": sum(group.peer_mapped(g) for g in range(n_gpus)),
                                "value": nq * args.steps / dtb, "unit": "queries/s", "ms_per_step": dtb / args.steps * 1e3,
                                "exchange_ms": tp["exchange_ms"], "merge_ms": tp["merge_ms"], "local_search_ms": tp["local_search_ms"]}
                group.set_exchange(group.EXCHANGE_RCCL)
                searcher.scan_timing(2)
        elif comm is not None:
            t = comm.last_timing()
            breakdown = {"scan_ms": scan_ms / max(scan_launches, 1), "local_search_ms": t["local_search_ms"],
                         "tail_ms": t["local_search_ms"] - scan_ms / max(scan_launches, 1), "exchange_ms": t["exchange_ms"],
                         "merge_ms": t["merge_ms"], "of": "last timed step on rank 0; exchange_ms includes waiting for the slowest rank"}
        else:
            breakdown = {"scan_ms": scan_ms / max(scan_launches, 1),
                         "tail_ms": elapsed / args.steps * 1e3 - scan_ms / max(scan_launches, 1), "exchange_ms": 0.0,
                         "of": "averages of the timed loop; tail = pack + tournament + exact re-score + certificate + host hand-off"}
    except Exception as e:  # noqa: BLE001
        breakdown = {"error": repr(e)}

    # second operating point, outside the headline's timed region: 128 queries per pass, where the scan is HBM-bound
    # (at 256 the same stream feeds twice the MFMA work and the pass is bound by the chip's power budget instead)
    alt = None
    if nq > 128:
        searcher.scan_timing(2)
        n_alt = max(3, args.steps // 2)
        sync_all()
        ta = time.perf_counter()
        for i in range(n_alt):
            qptr = qsets.device_ptr + (i % n_batches) * nq * D * 2
            if in_process:
                group.bruteforce_topk_dev(qptr, 128, k, out_s.data_ptr(), out_i.data_ptr(), mse.MODE_MFMA)
            elif host_exchange is not None:
                host_step(qptr, 128, mse.MODE_MFMA, out_s, out_i)
            elif comm is not None:
                comm.search_dev(searcher, qptr, 128, k, out_s.data_ptr(), out_i.data_ptr(), mse.MODE_MFMA, id_offset=lo)
            else:
                searcher.bruteforce_topk_dev(qptr, 128, k, out_s.data_ptr(), out_i.data_ptr(), mse.MODE_MFMA, id_offset=lo)
        sync_all()
        dta = time.perf_counter() - ta
        a_ms, a_n = searcher.scan_timing(0)
        alt = {"queries_per_step": 128, "value": 128 * n_alt / dta, "unit": "queries/s", "ms_per_step": dta / n_alt * 1e3, "steps": n_alt,
               "avg_launch_ms": a_ms / max(a_n, 1)}
        step(args.warmup + args.steps - 1)   # restore the last headline step's answer for the check below
        sync_all()

    # correctness spot check outside the timed region: the batched (MFMA) answer of the last step must equal the
    # exact-order kernel's answer for the same queries (two independent kernels), over the whole (sharded) index
    last = (args.warmup + args.steps - 1) % n_batches
    qptr = qsets.device_ptr + last * nq * D * 2
    m = min(8, nq)
    chk_s = torch.empty((m, k), dtype=torch.int64, device="cuda")
    chk_i = torch.empty((m, k), dtype=torch.int32, device="cuda")
    if in_process:
        group.bruteforce_topk_dev(qptr, m, k, chk_s.data_ptr(), chk_i.data_ptr(), mse.MODE_EXACT)
    elif host_exchange is not None:
        host_step(qptr, m, mse.MODE_EXACT, chk_s, chk_i)
    elif comm is not None:
        comm.search_dev(searcher, qptr, m, k, chk_s.data_ptr(), chk_i.data_ptr(), mse.MODE_EXACT, id_offset=lo)
    else:
        searcher.bruteforce_topk_dev(qptr, m, k, chk_s.data_ptr(), chk_i.data_ptr(), mse.MODE_EXACT, id_offset=lo)
    sync_all()
    verified = bool(torch.equal(chk_s, out_s[:m]) and torch.equal(chk_i, out_i[:m]))

    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # --gpus N > 1: the approximate-search paths over the same kind of shards (rows AND their codes / graphs partitioned), same exchange
    sharded_ann = None
    if n_gpus > 1 and not args.no_sharded_ann:
        import bench_ann
        try:
            with _stdout_to_stderr():
                if in_process:
                    sharded_ann = bench_ann.sharded_ann_inprocess(n_gpus, n_dev, int(args.ann_rows_per_gpu))
                elif comm is not None:
                    sharded_ann = bench_ann.sharded_ann_rank(comm, dist, rank, world, int(args.ann_rows_per_gpu))
                else:
                    sharded_ann = {"skipped": "RCCL did not come up (the brute-force line fell back to a gloo exchange)"}
        except Exception as e:  # noqa: BLE001
            sharded_ann = {"error": repr(e)}
        ffi.check(ffi.lib().mse_set_device(local_rank), "mse_set_device")

    # ---- side legs: each under the command's time budget (a leg that would not fit what is left is skipped and says so), a failure
    # is reported in the leg's object, never fatal; seconds per leg go to the detail file ----
    leg_seconds = {}
    reserve = 0.0 if (args.no_cpu_baseline or n_gpus > 1) else 45.0      # the CPU baseline still comes after the legs

    def run_leg(name, expect_s, fn, cond=True):
        if not cond:
            return None
        left = float(args.time_budget) - (time.perf_counter() - T_START) - reserve
        if expect_s > left:
            print(f"[bench] leg {name} skipped: {left:.0f} s left of the time budget, needs about {expect_s:.0f}", file=sys.stderr)
            return {"skipped": f"time budget: {left:.0f} s left of {float(args.time_budget):.0f}, the leg needs about {expect_s:.0f} s"}
        t_leg = time.perf_counter()
        try:
            r = fn()
        except Exception as e:  # noqa: BLE001
            r = {"error": repr(e)}
        leg_seconds[name] = round(time.perf_counter() - t_leg, 1)
        print(f"[bench] leg {name}: {leg_seconds[name]} s (t = {time.perf_counter() - T_START:.0f} s)", file=sys.stderr)
        return r

    leg_seconds["setup_and_headline"] = round(time.perf_counter() - T_START, 1)
    one = rank == 0 and n_gpus == 1 and world == 1
    callers_line = run_leg("concurrent_callers", 15, lambda: concurrent_callers_bench(vecs, k, nq * args.steps / elapsed), one and not args.no_callers)
    index_line = run_leg("index_callers_1e5", 10, lambda: index_callers_bench(k), one and not args.no_callers)
    shard_line = run_leg("shard_point", 10, lambda: shard_point_bench(k, nq, elapsed / args.steps * 1e3), one and not args.no_shard_point)

    # ---- second half of BASELINE.json's metric: SigLIP image embeds/s/GPU (replicas, no collective) ----
    siglip_line = None
    if not args.no_siglip:
        if world > 1:
            siglip_line = siglip_bench(args, world, rank, dist)       # every rank takes part (a barrier inside): no per-rank skipping
        else:
            siglip_line = run_leg("siglip", 45, lambda: siglip_bench(args, world, rank, dist))
    pq_line = run_leg("pq_scan", 25, lambda: pq_bench(args), rank == 0 and n_gpus == 1 and not args.no_pq)
    graph_line = run_leg("graph_search", 15, lambda: graph_bench(args), rank == 0 and n_gpus == 1 and not args.no_graph)

    gscale_line = ann_line = g1e8_line = request_line = None
    if rank == 0 and n_gpus == 1 and not (args.no_graph_scale and args.no_ann_scale):
        del searcher, vecs                             # the 230 GB index makes room for the clustered sets of the next two legs
        import gc
        gc.collect()
    if one and not args.no_ann_scale:
        ann_line = run_leg("ann_1e8", 30, lambda: ann_scale_bench(args))
        import gc
        gc.collect()
        torch.cuda.empty_cache()
    if rank == 0 and n_gpus == 1 and not args.no_graph_scale:
        # one row per synthetic set (bench_ann.py): easy (rounds 1-4), hard (no micro-clusters), ood (queries from another distribution,
        # graph built with a query sample + robust_stitch); the request path's call shape (T threads x 1 query) on the hard set
        import gc
        import bench_ann
        gscale_line = {"metric": f"queries/sec over a {int(args.graph_scale_rows):.0e}x1152 graph index @ recall@10>=0.95 (GPU-resident beam search)", "unit": "queries/s", "sets": {}}
        expect = {"hard": 125.0, "easy": 80.0, "ood": 125.0}
        scale_rows = float(args.graph_scale_rows) / 1e7
        for kind in [x for x in args.graph_kinds.split(",") if x]:
            gscale_line["sets"][kind] = run_leg(
                "graph_index_" + kind, expect.get(kind, 100.0) * scale_rows,
                lambda: bench_ann.graph_index_bench(ROOT, kind, int(args.graph_scale_rows), batch=int(args.graph_batch), passes=int(args.graph_passes),
                                                    callers=(kind in ("easy", "hard") and not args.no_callers),
                                                    request_path=(kind == "hard" and not args.no_siglip and not args.no_request_path)))
            gc.collect()
            torch.cuda.empty_cache()
        request_line = (gscale_line["sets"].get("hard") or {}).pop("request_path", None) if isinstance(gscale_line["sets"].get("hard"), dict) else None
        if args.graph_1e8 and world == 1:
            rate = ((gscale_line["sets"].get("easy") or {}).get("build") or {}).get("points_per_s")
            try:
                left = float(args.time_budget) - (time.perf_counter() - T_START) - 90.0     # the CPU baseline and the tail of the line still come
                t_leg = time.perf_counter()
                g1e8_line = bench_ann.graph_index_1e8(ROOT, rate, min(float(args.graph_1e8_budget), left))
                leg_seconds["graph_index_1e8"] = round(time.perf_counter() - t_leg, 1)
            except Exception as e:  # noqa: BLE001
                g1e8_line = {"error": repr(e)}
            gc.collect()
            torch.cuda.empty_cache()
        head = (gscale_line["sets"].get("hard") or {}).get("exact_scored", {}).get("held_out") or {}
        gscale_line.update({"value": head.get("queries_per_s"), "recall_at_10": head.get("recall_at_10"), "search_list": head.get("value"),
                            "value_is": "the HARD set, exactly scored neighbours, one call of 4096 held-out queries"})

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        qps = nq * args.steps / elapsed
        avg_scan_ms = scan_ms / max(scan_launches, 1)
        bytes_per_launch = (hi - lo) * D * 2
        achieved = bytes_per_launch / (avg_scan_ms * 1e-3) / 1e9 if scan_launches else None
        mfma_tflops = 2.0 * (hi - lo) * D * min(nq, tile) / (avg_scan_ms * 1e-3) / 1e12 if scan_launches else None
        if alt:
            alt["roofline"] = {"bound": "hbm", "achieved": bytes_per_launch / (alt["avg_launch_ms"] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": bytes_per_launch / (alt["avg_launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS}
        # HBM traffic from the PMC passes (collected offline, one counter per rocprofv3 run -- it cannot be read
        # inside a timed run); only reported when this run is the profiled configuration
        traffic = None
        try:
            pm = json.load(open(PMC_TRAFFIC))
            pp = pm["per_pass"].get(str(min(nq, tile))) if pm["rows"] == hi - lo else None
            if pp:
                traffic = pp["hbm_read_bytes_per_launch"] + pp["hbm_write_bytes_per_launch"]
        except Exception:
            traffic = None
        line = {
            "metric": "queries/sec over 1e8x1152 index @ recall@10>=0.95 (exact brute force: recall 1.0)",
            "value": qps,
            "unit": "queries/s",
            "n_gpus": n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f16 in, f32 accumulate, i64 fixed-point scores",
            "data": "synthetic (device-generated unit-norm rows, seeds 0x5EED0001/2)",
            "config": {"workload": f"brute-force top-{k} over {n_total} x {D} fp16 rows, {nq} queries/step, "
                                   f"row-sharded over {n_gpus} GPU(s)" + (" + FALLBACK gloo all-gather of [Q,k] records (RCCL did not come up)" if host_exchange is not None else
                                                                          " + RCCL all-gather of [Q,k] records" if world > 1 else
                                                                          " + RCCL all-gather of [Q,k] records (one process, a thread per GPU)" if in_process and exchange.get("rccl_ranks") else
                                                                          " + peer-mapped gather of [Q,k] records" if in_process else ""),
                       "rows_total": n_total, "rows_per_gpu": hi - lo, "queries_per_step": nq, "queries_per_step_pick": queries_pick, "k": k,
                       "parallelism": f"row-shard x{n_gpus}", "exchange": exchange},
            "roofline": {"bound": "hbm", "kernel": {320: "scan_mfma_kernel<2,20>", 256: "scan_mfma2d_kernel<3,16>", 192: "scan_mfma_kernel<3,12>", 128: "scan_mfma_kernel<3,8>"}.get(tile, "scan_mfma") + f" ({nq} queries per pass)", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": (achieved / HBM_PEAK_GBS) if achieved else None,
                         "traffic": traffic, "traffic_source": ("profiles/" + os.path.basename(PMC_TRAFFIC) + " (rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE)") if traffic else None,
                         "bytes_per_launch": bytes_per_launch, "avg_launch_ms": avg_scan_ms, "queries_per_launch": min(nq, tile),
                         # the same launch against the matrix cores: 2*rows*1152*queries flops; dense f16 peak 2500 TFLOP/s
                         "mfma_tflops": mfma_tflops, "mfma_frac": (mfma_tflops / 2500.0) if mfma_tflops else None,
                         "note": "256 / 320 queries per pass: neither HBM nor the matrix cores are saturated; the pass is bound by the "
                                 "power budget (rocm-smi beside it: 1370 W of the 1400 W board limit, engine clock 1.5-1.6 GHz of 2.4 -- "
                                 "profiles/r04_scan_variants.txt; every 64 more queries per pass cost 9-11 ms on top of the 40 ms stream; "
                                 "profiles/r02_power_clocks.txt; the same kernel on all-zero rows is 17-20 % faster; tilings with a third "
                                 "fewer LDS reads, deeper prefetch or no barrier at all take the same 8.9 M cycles per 1e7 rows -- "
                                 "profiles/r03_scan_variants.txt, DESIGN.md 3.1).  hbm_bound_point = the 128-query pass (HBM-bound).",
                         "launches_timed": scan_launches,
                         # informational: the guide's measured float4-copy ceiling of this chip is 6.29 TB/s
                         "frac_of_measured_copy_ceiling": (achieved / 6290.0) if achieved else None},
            "step_breakdown": breakdown,
            "exchange_alt": exchange_alt,
            "verified_vs_exact_kernel": verified,
            "hbm_bound_point": alt,
            "certificate": stats,
        }
        if callers_line:
            line["concurrent_callers"] = callers_line
        if index_line:
            line["index_callers_1e5"] = index_line
        if shard_line:
            line["shard_point"] = shard_line
        if siglip_line:
            line["siglip"] = siglip_line
        if pq_line:
            line["pq_scan"] = pq_line
        if graph_line:
            line["graph_search"] = graph_line
        if ann_line:
            line["ann_1e8"] = ann_line
        if gscale_line:
            line["graph_index_1e7"] = gscale_line
        if request_line:
            line["request_path"] = request_line
        if g1e8_line:
            line["graph_index_1e8"] = g1e8_line
        if sharded_ann:
            line["sharded_ann"] = sharded_ann
        if note:
            line["note"] = note
        if n_gpus == 1 and not args.no_cpu_baseline:
            t_leg = time.perf_counter()
            line["cpu_baseline"] = cpu_baseline(n_total, k)
            if graph_line and "error" not in graph_line and "skipped" not in graph_line:
                line["cpu_baseline"]["graph_build"] = cpu_graph_build(int(args.graph_rows))
            leg_seconds["cpu_baseline"] = round(time.perf_counter() - t_leg, 1)
        leg_seconds["total"] = round(time.perf_counter() - T_START, 1)
        line["leg_seconds"] = leg_seconds
        emit(line, args.detail)
    if group is not None:
        group.close()
    if comm is not None:
        comm.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
