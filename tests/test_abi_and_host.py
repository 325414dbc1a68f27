"""CPU-side checks of the drop-in boundary: the shared library loads, exports every symbol that
include/mse.h declares, the host-only entry points behave like the reference, and the compute
entry points fail loudly (no CPU fallback) when there is no device."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, ROOT


def declared_functions():
    text = open(os.path.join(ROOT, "include", "mse.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mse_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound(mse):
    from mse import ffi
    names = declared_functions()
    assert len(names) > 50
    L = C.CDLL(ffi.LIB_PATH)
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, f"declared in include/mse.h but not exported: {missing}"
    unbound = [n for n in names if n not in ffi.SIGNATURES]
    assert not unbound, f"declared but not bound in mse/ffi.py: {unbound}"
    extra = [n for n in ffi.SIGNATURES if n not in names]
    assert not extra, f"bound but not declared: {extra}"


def test_only_mse_symbols_are_public(mse):
    import subprocess
    from mse import ffi
    out = subprocess.check_output(["nm", "-D", "--defined-only", ffi.LIB_PATH], text=True)
    public = [l.split()[-1] for l in out.splitlines() if " T " in l]
    assert all(p.startswith("mse_") or p in ("_init", "_fini") for p in public), public


def test_scale_dot_matches_oracle(mse, orc):
    for v in (0.0, 1.0, -1.0, 0.5, 1e-10, -3e-10, 1e30, -1e30, float("nan"), float("inf"), 0.123456789):
        assert mse.scale_dot_result(v) == orc.lib().orc_scale_dot_result(v)
        assert mse.scale_dot_result_f64(v) == orc.lib().orc_scale_dot_result_f64(v)


def test_neighbour_buffer_matches_oracle_trace(mse, orc):
    g = np.load(os.path.join(GOLDEN, "neighbour_buffer_trace.npz"))
    nb = mse.NeighbourBuffer(int(g["cap"]))
    assert nb.cap() == int(g["cap"])
    for step, (op, a, b) in enumerate(g["ops"]):
        if op == 0:
            nb.insert(int(a), int(b))
        else:
            r = nb.next_unvisited()
            assert (-1 if r is None else r) == int(a)
        n = int(g["lens"][step])
        assert len(nb) == n
        assert np.array_equal(nb.ids, g["ids"][step, :n])
        assert np.array_equal(nb.scores, g["scores"][step, :n])
    nb.clear()
    assert len(nb) == 0 and nb.next_unvisited() is None


def test_neighbour_buffer_random_against_oracle(mse, orc):
    rng = np.random.default_rng(9)
    for cap in (1, 2, 7, 64):
        a, b = mse.NeighbourBuffer(cap), orc.NeighbourBuffer(cap)
        for _ in range(500):
            if rng.random() < 0.75:
                i, s = int(rng.integers(0, 40)), int(rng.integers(-5, 5)) << 20
                a.insert(i, s)
                b.insert(i, s)
            else:
                assert a.next_unvisited() == b.next_unvisited()
            assert np.array_equal(a.ids, b.ids) and np.array_equal(a.scores, b.scores)


def test_descriptor_product_matches_oracle(mse, orc):
    rng = np.random.default_rng(10)
    desc = rng.integers(0, 256, size=(50, 4), dtype=np.uint8)
    scales = (rng.standard_normal(4) / 512).astype(np.float32)
    for i in (0, 7, 49):
        assert mse.descriptor_product(scales, desc, i) == orc.descriptor_product(scales, desc, i)


def test_get_total_embedding(mse, orc):
    rng = np.random.default_rng(12)
    d = 64
    embs = orc.f16_bits(rng.standard_normal((3, d)).astype(np.float32))
    calls = []

    def server(batch):
        calls.append(batch)
        if "images" in batch:
            return [embs[0].tobytes()]
        return [embs[1].tobytes(), embs[2].tobytes()]

    raw = rng.standard_normal(d).astype(np.float32)
    terms = [{"image": b"bmpbytes", "weight": 2.0}, {"text": "cat"}, {"text": "dog", "weight": -0.5},
             {"embedding": raw.tolist(), "weight": 0.25}, {"predefined_embedding": "nsfw", "weight": 1.5},
             {"predefined_embedding": "missing"}]
    pre = {"nsfw": rng.standard_normal(d).astype(np.float32)}
    got = mse.get_total_embedding(terms, d, server, pre)
    # images batch first, then text (common.rs:252-266)
    assert list(calls[0]) == ["images"] and calls[1] == {"text": ["cat", "dog"]}
    want = raw * np.float32(0.25) + pre["nsfw"] * np.float32(1.5)
    want = want + orc.total_embedding(embs, np.array([2.0, 1.0, -0.5], np.float32))
    assert np.allclose(got, want, rtol=1e-6, atol=1e-6)
    assert mse.decode_fp16_buffer(embs[0].tobytes()).dtype == np.float32
    assert np.array_equal(mse.chunk_fp16_buffer(embs[0].tobytes()), embs[0])


def test_shape_violations_are_errors_not_ub(mse):
    from mse import ffi
    L = ffi.lib()
    # widths must be multiples of 64 (fast_dot debug_assert, vector.rs:197,259)
    assert not L.mse_base_wrap_device(None, 10, 100)
    assert "multiple of 64" in ffi.last_error()
    assert not L.mse_index_new(100)
    # more than 256 centroids (vector.rs:337)
    c = np.zeros((257, 64), np.float32)
    t = np.zeros((64, 64), np.float32)
    assert not L.mse_pq_load(c.ctypes.data_as(ffi.f32p), 257, t.ctypes.data_as(ffi.f32p), 64, 16)
    assert "256" in ffi.last_error()
    with pytest.raises(mse.MseError):
        mse.ProductQuantizer(np.zeros((4, 64), np.float32), np.zeros(10, np.float32), 16, 64)   # transform.len() != d*d (:334)
    with pytest.raises(mse.MseError):
        mse.VectorList.from_f16s(np.zeros(100, np.uint16), 64)                                     # assert at :174
    with pytest.raises(mse.MseError):
        mse.fast_dot(np.zeros(64, np.uint16), np.zeros(128, np.uint16))


def test_compute_fails_loudly_without_a_device(mse):
    from mse import ffi
    if ffi.lib().mse_device_count() > 0:
        pytest.skip("a device is present")
    with pytest.raises(mse.MseError):
        mse.VectorList.from_f16s(np.zeros((4, 64), np.uint16), 64)
    with pytest.raises(mse.MseError):
        mse.fast_dot(np.zeros(64, np.uint16), np.zeros(64, np.uint16))
    with pytest.raises(mse.MseError):
        mse.ScalarQuantizerIndex(64).add(np.zeros((1, 64), np.float32))


def test_product_package_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "meme-search-engine_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import orc" not in text and "liboracle" not in text and "from oracle" not in text, (dirpath, f)


def test_shard_ranges_and_merge(mse, orc):
    from mse import shard
    for n, w in ((10, 3), (100000000, 8), (7, 8), (0, 2)):
        spans = [shard.shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
        assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
    import torch
    rng = np.random.default_rng(13)
    scores = rng.integers(-5, 5, size=(6, 24)).astype(np.int64)
    ids = rng.permutation(10000)[:6 * 24].reshape(6, 24).astype(np.uint32)
    ids[2, 5:9] = 0xFFFFFFFF
    s_np, i_np = shard.merge_topk_numpy(scores, ids, 10)
    s_t, i_t = shard.merge_topk_torch(torch.from_numpy(scores), torch.from_numpy(ids.astype(np.int64)), 10)
    assert np.array_equal(s_np, s_t.numpy()) and np.array_equal(i_np, i_t.numpy().astype(np.uint32))
    for q in range(6):
        valid = ids[q] != 0xFFFFFFFF
        order = np.lexsort((ids[q][valid], -scores[q][valid]))[:10]
        assert np.array_equal(i_np[q], ids[q][valid][order])


def test_topk_of_visited_sorts_by_exact_score(mse):
    """The server's last step (src/query_disk_index.rs:529-540) over the padded arrays of a batched search."""
    res = {"visited_ids": np.array([[5, 6, 7, 9], [1, 2, 0, 0], [4, 0, 0, 0]], np.uint32),
           "visited_scores": np.array([[10, 30, 30, 99], [5, -7, 0, 0], [np.iinfo(np.int64).max, 0, 0, 0]], np.int64),
           "n_visited": np.array([3, 2, 1], np.uint32)}
    top = mse.topk_of_visited(res, 3)
    assert top[0].tolist() == [6, 7, 5]                       # equal scores keep visit order; the fourth column is past n_visited
    assert top[1].tolist() == [1, 2, mse.ID_NONE]
    assert top[2].tolist() == [4, mse.ID_NONE, mse.ID_NONE]


def test_bench_launch_shapes_are_checked_before_any_device_work():
    """bench.py --gpus N runs bare (one process over N devices) or under torchrun (WORLD_SIZE = N): the two ways of getting that
    wrong are reported at once, with no device needed."""
    import subprocess
    import sys
    bench = os.path.join(ROOT, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, bench, "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "HIP device(s) visible" in (r.stderr + r.stdout)
    r = subprocess.run([sys.executable, bench, "--gpus", "2"], env=dict(env, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=4" in (r.stderr + r.stdout)


def test_bench_line_stays_small_whatever_the_legs_return():
    """The driver parses the LAST stdout line of bench.py into its record: round 5's 40 KB line came back unparsed.  The printed line is
    bench_line.compact_line(full) -- contract scalars, config, roofline with FLAT per-leg scalars, cpu_baseline -- and stays under 4 KB
    for round 5's full result (a committed file), for a result whose every string is huge and for one with no legs at all."""
    import json
    import bench_line
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_default.json")))
    line = bench_line.compact_line(full)
    text = json.dumps(line)
    assert len(text) < 4096
    assert json.loads(text) == line
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert key in line, key
    assert line["config"]["workload"].startswith("brute-force top-10 over 100000000 x 1152") and line["config"]["queries_per_step"] == 320
    roof = line["roofline"]
    assert roof["bound"] == "hbm" and roof["peak"] == 8000.0 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3 and roof["traffic"] > 2.3e11
    assert all(not isinstance(v, (dict, list)) for v in roof["legs"].values())          # flat scalars only
    assert roof["legs"]["siglip_img_s"] > 1000 and roof["legs"]["graph_hard_recall"] > 0.95 and roof["legs"]["pq_frac"] > 0.5
    assert set(line["cpu_baseline"]) == {"value", "unit", "cores", "kind", "sample"} and line["cpu_baseline"]["kind"] == "port"
    # a hostile full result: long strings everywhere, a hundred synthetic sets
    fat = json.loads(json.dumps(full))
    fat["config"]["workload"] = "w" * 5000
    fat["config"]["exchange"] = {"kind": "k" * 5000, "rccl_ranks": 8, "rccl_unavailable": "u" * 5000}
    fat["note"] = "n" * 5000
    fat["cpu_baseline"]["sample"] = "s" * 5000
    fat["roofline"]["kernel"] = "r" * 5000
    fat["graph_index_1e7"]["sets"] = {f"set{i:03d}": fat["graph_index_1e7"]["sets"]["hard"] for i in range(100)}
    text = json.dumps(bench_line.compact_line(fat))
    assert len(text) <= 4096 and json.loads(text)["value"] == line["value"] and "cpu_baseline" in json.loads(text)
    # nothing but the contract
    bare = bench_line.compact_line({"metric": "m", "value": 1.0, "unit": "u", "n_gpus": 1, "steps": 1, "warmup": 0})
    assert bare["roofline"]["legs"] == {} and bare["value"] == 1.0
    # legs that failed or were skipped carry no scalars and do not break the line
    broken = dict(full, siglip={"error": "x"}, pq_scan={"skipped": "time budget"}, graph_index_1e7={"sets": {"hard": {"error": "y"}, "easy": None}})
    legs = bench_line.compact_line(broken)["roofline"]["legs"]
    assert "siglip_img_s" not in legs and "pq_frac" not in legs and "graph_hard_qps" not in legs and "hbm128_frac" in legs


def test_bench_dry_run_prints_the_small_line_for_every_launch_shape():
    """`bench.py --dry-run` needs no device: it checks the launch shape exactly as a real run does and prints the compact line, so the
    commands the driver uses for 1 / 2 / 4 / 8 GPUs (bare and under torchrun) cannot die on a flag."""
    import json
    import subprocess
    import sys
    bench = os.path.join(ROOT, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    for n in (1, 2, 4, 8):
        for launch in ("bare", "torchrun"):
            e = dict(env, WORLD_SIZE=str(n), RANK="0", LOCAL_RANK="0") if launch == "torchrun" else env
            extra = ["--logical-shards"] if n == 2 else []
            r = subprocess.run([sys.executable, bench, "--gpus", str(n), "--steps", "7", "--warmup", "2", "--dry-run"] + extra, env=e,
                               capture_output=True, text=True, timeout=120)
            assert r.returncode == 0, r.stderr
            last = r.stdout.strip().splitlines()[-1]
            assert len(last) < 4096
            line = json.loads(last)
            assert line["n_gpus"] == n and line["steps"] == 7 and line["warmup"] == 2 and line["value"] is None and "dry-run" in line["data"]
            assert line["config"]["rows_per_gpu"] == (100_000_000 + n - 1) // n and line["config"]["parallelism"] == f"row-shard x{n}"
            if n > 1:
                assert "rccl_ranks" in line["config"]["exchange"]
    # a rank other than 0 prints nothing; a world that disagrees with --gpus is an error
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--dry-run"], env=dict(env, WORLD_SIZE="2", RANK="1", LOCAL_RANK="1"), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == ""
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--dry-run"], env=dict(env, WORLD_SIZE="4", RANK="0"), capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stderr
    # the 1e8-row graph leg is opt-in; the old flag is still accepted
    r = subprocess.run([sys.executable, bench, "--dry-run", "--no-graph-1e8"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0


def test_bench_codec_trainer_lowers_the_query_aware_loss():
    """bench_ann.train_codec_aopq (bench data, not product: the 64 x 256 codec trained the way diskann/aopq_train.py:33-85 does -- Adam on
    the centroids against E_q[(q . residual)^2] under the codec's max-inner-product assignment, SVD rotation updates) on a small
    low-rank sample on the CPU: the loss it reports falls, the transform stays orthogonal, and the ADC error it leaves on the
    training queries is below that of its own starting point (random rotation + k-means)."""
    import sys
    import torch
    sys.path.insert(0, ROOT)
    import bench_ann as ba
    torch.manual_seed(0)
    n, d = 1500, ba.D
    basis = torch.randn(48, d) * (torch.arange(1, 49).float() ** -0.6).unsqueeze(1)
    x = torch.randn(n, 48) @ basis + 0.3 * torch.randn(n, d) / d ** 0.5 + 0.5 * torch.randn(1, d)
    x = x / x.norm(dim=1, keepdim=True)
    q = x[torch.randperm(n)[:300]] + 0.05 * torch.randn(300, d) / d ** 0.5
    q = q / q.norm(dim=1, keepdim=True)

    def adc_error(cents, T):
        P = torch.from_numpy(T.T.copy())
        xr, qr = x @ P, q @ P
        cent = torch.from_numpy(cents.reshape(256, 64, 18).transpose(1, 0, 2).copy())
        a = torch.einsum("nsd,skd->nsk", xr.view(-1, 64, 18), cent).argmax(2)
        y = cent[torch.arange(64).unsqueeze(0), a].reshape(-1, d)
        return float(((qr @ (xr - y).T) ** 2).mean())
    c0, T0 = ba.train_codec(x[:1000].numpy(), seed=4)
    c1, T1, info = ba.train_codec_aopq(x, q, rounds=2, iters=12, lr=1e-3, kmeans_iters=2)
    loss = info["query_aware_loss_first_last_per_round"]
    assert len(loss) == 4 and loss[1] < loss[0] and loss[3] < loss[0]
    assert np.allclose(T1 @ T1.T, np.eye(d), atol=2e-3)
    assert c1.shape == (256, d) and T1.shape == (d, d) and c1.dtype == np.float32
    assert adc_error(c1, T1) < 0.8 * adc_error(c0, T0)


DEV_KNOBS = ["MSE_SCAN_ABL", "MSE_SCAN_2D", "MSE_SCAN_S", "MSE_ATT_ABL", "MSE_ATT64_ABL", "MSE_ATT_WAVES", "MSE_ATT_QT", "MSE_ATT_TILE32",
             "MSE_GEMM_RANDOM", "MSE_GEMM_OLD256", "MSE_GEMM_128", "MSE_GEMM_NOPERSIST", "MSE_GEMM_STAGGER", "MSE_GEMM_NONARROW",
             "MSE_PQ_OLDTRANSFORM", "MSE_PQ_OLDQUANT", "MSE_PQ_OLDSCAN", "MSE_DEDUP_OLD"]
# what the product build may still read from the environment: hooks under which every answer stays correct
PRODUCT_HOOKS = {"MSE_BUILD_EXACT_BACKEDGE", "MSE_BUILD_EXACT_PRUNE", "MSE_GRAM_EPS_SCALE", "MSE_SHARD_NO_PEER", "MSE_SIGLIP_NOFUSE", "MSE_SIGLIP_NOSMALL", "MSE_COALESCE_NO_HOLD", "MSE_SIGLIP_TEXT_PARTS",
                 "MSE_SIGLIP_STREAMS", "MSE_BEAM_WAVES", "MSE_VISITED_BUDGET_KB", "MSE_VISITED_MODE", "MSE_VISITED_TABLE_BITS"}


def test_product_library_reads_no_developer_knob(mse):
    """Timing ablations (wrong answers by design) and superseded kernels are compiled only into the developer library
    (make dev, -DMSE_DEV_KERNELS).  The product .so must not even contain the names of those environment variables, and the
    only MSE_* names it does contain are the answer-preserving hooks."""
    import re
    from mse import ffi
    blob = open(ffi.LIB_PATH, "rb").read()
    names = {m.decode() for m in re.findall(rb"MSE_[A-Z0-9_]{3,}", blob)}
    names = {n for n in names if not n.startswith(("MSE_HIP_TRY", "MSE_DYN_LDS", "MSE_DEV_K", "MSE_ID_", "MSE_API"))}
    assert not names & set(DEV_KNOBS), names & set(DEV_KNOBS)
    assert names <= PRODUCT_HOOKS, names - PRODUCT_HOOKS


@pytest.mark.parametrize("threads,rounds,max_queries", [(1, 50, 256), (64, 40, 256), (200, 20, 32), (16, 200, 4)])
def test_coalescer_queue_hands_every_caller_its_own_answer(mse, threads, rounds, max_queries):
    """The cross-thread coalescer's queue (csrc/dispatch.hip) without a device: T host threads x R one-query requests through a
    stand-in pass.  Every caller gets the answer for ITS payload, a failing request fails alone with its own message, nothing
    deadlocks; a lone caller is never batched with a wait (as many passes as requests), many callers share passes, and no pass
    carries more than `max_queries`."""
    import ctypes as C
    from mse import ffi
    stats = (C.c_uint64 * 6)()
    bad = C.c_uint64(12345)
    ffi.check(ffi.lib().mse_debug_coalescer_selftest(threads, rounds, max_queries, 2000, stats, C.byref(bad)))
    assert bad.value == 0
    queries, requests, passes, max_pass = int(stats[0]), int(stats[1]), int(stats[2]), int(stats[3])
    assert queries == requests == threads * rounds
    assert max_pass <= max_queries
    if threads == 1:
        assert passes == requests
    elif threads >= 64:
        assert passes < requests / 2, (passes, requests)


@pytest.mark.parametrize("async_threads,window,n,sync_threads,max_queries,workers,own_queues",
                         [(1, 1, 300, 0, 256, 1, 0), (1, 512, 6000, 0, 128, 2, 0), (3, 64, 2000, 8, 64, 2, 0), (2, 4096, 9000, 16, 1024, 2, 0),
                          (4, 128, 3000, 8, 256, 2, 1), (1, 1, 200, 0, 64, 1, 1)])
def test_coalescer_asynchronous_requests(mse, async_threads, window, n, sync_threads, max_queries, workers, own_queues):
    """submit_async / completions of the coalescer without a device: threads that keep a window of records in flight and collect
    whatever has completed (theirs or another thread's), beside blocking callers on the same handle.  Every record comes back exactly
    once with ITS answer or ITS error; the blocking callers are unaffected; a window of one is never batched with a wait.  With
    own_queues every asynchronous thread has a completion queue of its own (one per event loop of a host that runs several) and must
    get back exactly the records it submitted."""
    import ctypes as C
    from mse import ffi
    stats = (C.c_uint64 * 6)()
    bad = C.c_uint64(12345)
    ffi.check(ffi.lib().mse_debug_coalescer_selftest_async(async_threads, window, n, sync_threads, max_queries, workers, own_queues, stats, C.byref(bad)))
    assert bad.value == 0
    assert stats[5] == async_threads * n                       # every asynchronous record collected
    assert stats[0] == async_threads * n + sync_threads * 200  # and all of them, blocking ones included, executed
    assert stats[3] <= max_queries
    if window == 1 and async_threads == 1 and sync_threads == 0:
        assert stats[2] == n                                   # a lone request in flight: one pass each, no batching delay
    if window >= 512:
        assert stats[2] < stats[0] / 8                         # windows share passes


@pytest.mark.parametrize("threads,rounds,max_queries,workers", [(64, 40, 256, 2), (600, 10, 128, 2), (24, 100, 8, 3)])
def test_coalescer_with_several_workers(mse, threads, rounds, max_queries, workers):
    """The same queue with several worker threads (the graph's request path runs two) and its slotted completion (256 consecutive
    arrivals share a wake-up slot): every caller still gets its own answer, nothing deadlocks, no pass exceeds its size."""
    import ctypes as C
    from mse import ffi
    stats = (C.c_uint64 * 6)()
    bad = C.c_uint64(12345)
    ffi.check(ffi.lib().mse_debug_coalescer_selftest_workers(threads, rounds, max_queries, 2000, workers, stats, C.byref(bad)))
    assert bad.value == 0
    assert int(stats[0]) == int(stats[1]) == threads * rounds
    assert int(stats[3]) <= max_queries
    assert int(stats[2]) < threads * rounds / 2
