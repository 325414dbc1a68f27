"""bench_line.py -- the ONE line `bench.py` prints, kept small.

The driver parses the last stdout line of `bench.py` into its record; round 5's line had grown to 40 KB and the record came back
unparsed.  `compact_line(full)` reduces the full result object (every leg's detail, written to `bench_detail.json` and never to
stdout) to the contract's scalars, `config`, `roofline` (the dominant kernel + `legs`: FLAT scalars only, one number per key) and
`cpu_baseline` -- at most LINE_LIMIT bytes whatever the legs returned.  Pure Python, no device: tests/test_abi_and_host.py runs it over
a committed full result and over synthetic worst cases.
"""
import json
import math

LINE_LIMIT = 4096          # hard bound of the printed line (bytes of its JSON text)
LINE_TARGET = 3800         # what the trimming below aims for: margin for the float digits of another box

CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")


def g(o, *path):
    """o[path[0]][path[1]]... or None; list indices allowed."""
    for p in path:
        if isinstance(o, dict):
            o = o.get(p)
        elif isinstance(o, (list, tuple)) and isinstance(p, int) and -len(o) <= p < len(o):
            o = o[p]
        else:
            return None
    return o


def num(v, digits=4):
    """A JSON-safe scalar with few digits (4 significant by default), or None."""
    if isinstance(v, bool) or v is None:
        return v
    if isinstance(v, int):
        return v
    if isinstance(v, float):
        if not math.isfinite(v):
            return None
        if v == 0.0:
            return 0.0
        r = round(v, max(0, digits - 1 - int(math.floor(math.log10(abs(v))))))
        return int(r) if abs(r) >= 10 ** digits else r
    return None


def text(v, limit):
    return v if v is None else str(v)[:limit]


def flat_legs(full):
    """Every side leg's headline numbers, one scalar per key.  Order = importance: the trimming drops from the END."""
    L = {}

    def put(key, v, digits=4):
        v = num(v, digits)
        if v is not None:
            L[key] = v

    put("hbm128_frac", g(full, "hbm_bound_point", "roofline", "frac"))
    put("hbm128_qps", g(full, "hbm_bound_point", "value"))
    sg = full.get("siglip") or {}
    put("siglip_img_s", sg.get("value"))
    put("siglip_frac", g(sg, "roofline", "frac"))
    put("text_s", g(sg, "text_tower", "value"))
    put("text_frac", g(sg, "text_tower", "roofline", "frac"))
    put("server_img_s", sg.get("server_images_per_s"))
    put("text1_ms", g(sg, "latency_ms", "text", "1"))
    put("image1_ms", g(sg, "latency_ms", "image", "1"))
    put("text8_ms", g(sg, "latency_ms", "text", "8"))
    put("text32_ms", g(sg, "latency_ms", "text", "32"))
    pq = full.get("pq_scan") or {}
    put("pq_frac", g(pq, "roofline", "frac"))
    put("pq_burst_frac", g(pq, "roofline", "burst", "frac"))
    put("pq_e2e_frac", g(pq, "roofline", "end_to_end", "frac"))
    put("pq_qps", pq.get("queries_per_s_batched"))
    for kind, row in (g(full, "graph_index_1e7", "sets") or {}).items():
        if not isinstance(row, dict):
            continue
        h = g(row, "exact_scored", "held_out") or {}
        put(f"graph_{kind}_qps", h.get("queries_per_s"))
        put(f"graph_{kind}_recall", h.get("recall_at_10"))
        put(f"graph_{kind}_L", h.get("value"))
        rf = row.get("gather_roofline") or {}
        put(f"graph_{kind}_gather_GBps", rf.get("achieved"))
        put(f"graph_{kind}_frac", rf.get("frac"))
        for path, short in (("adc_scored", "adc"), ("pq_rerank", "rerank")):
            hh = g(row, path, "held_out") or {}
            put(f"graph_{kind}_{short}_qps", hh.get("queries_per_s"))
            put(f"graph_{kind}_{short}_recall", hh.get("recall_at_10"))
        put(f"graph_{kind}_pq_only_recall", row.get("pq_only_recall_at_10"))
        put(f"graph_{kind}_pq_only_recall_untrained", row.get("pq_only_recall_at_10_untrained_codec"))
        put(f"graph_{kind}_build_s", g(row, "build", "seconds"))
        for p in g(row, "graph_callers", "points") or []:
            if p.get("threads") == 4096:
                put(f"graph_{kind}_threads4096_qps", p.get("queries_per_s"))
        for p in g(row, "graph_callers", "tickets", "points") or []:
            if p.get("in_flight") == 4096 and p.get("host_threads", 1) == 1 and p.get("query_copied_at_submit", True):
                put(f"graph_{kind}_tickets4096_qps", p.get("queries_per_s"))
    rq = full.get("request_path") or {}
    put("request_text1_p50_ms", g(rq, "one_at_a_time", "latency_ms", "p50"))
    put("request_text1_p99_ms", g(rq, "one_at_a_time", "latency_ms", "p99"))
    put("request_64_qps", g(rq, "in_flight_64", "requests_per_s"))
    put("request_64_p50_ms", g(rq, "in_flight_64", "latency_ms", "p50"))
    put("request_64_p99_ms", g(rq, "in_flight_64", "latency_ms", "p99"))
    g8 = full.get("graph_index_1e8") or {}
    put("graph1e8_qps", g8.get("value"))
    put("graph1e8_recall", g8.get("recall_at_10"))
    put("graph1e8_L", g8.get("search_list"))
    put("graph1e8_build_s", g(g8, "build", "seconds"))
    cc = full.get("concurrent_callers") or {}
    put("callers512_qps", cc.get("queries_per_s"))
    put("callers512_vs_headline", cc.get("vs_resident_batch_headline"))
    sp = full.get("shard_point") or {}
    put("shard_ms", sp.get("ms_per_step"))
    put("shard_frac", g(sp, "roofline", "frac"))
    put("proj8_eff", g(sp, "projected_8gpu", "efficiency_vs_one_gpu_1e8"))
    put("ann1e8_qps", g(full, "ann_1e8", "queries_per_s"))
    put("ann1e8_recall", g(full, "ann_1e8", "recall_at_10"))
    put("index1e5_qps", g(full, "index_callers_1e5", "queries_per_s"))
    sa = full.get("sharded_ann") or {}
    put("sharded_pq_qps", g(sa, "pq_scan_rerank", "queries_per_s"))
    put("sharded_graph_qps", g(sa, "graph_index", "queries_per_s"))
    for key, v in (("sharded_pq_equal", g(sa, "pq_scan_rerank", "equals_the_unsharded_call_bit_for_bit")),
                   ("sharded_graph_equal", g(sa, "graph_index", "equals_the_merge_of_per_shard_calls"))):
        if isinstance(v, bool):
            L[key] = v
    if isinstance(sa, dict) and (sa.get("error") or sa.get("skipped")):
        L["sharded_ann_note"] = text(sa.get("error") or sa.get("skipped"), 80)
    put("exchange_alt_qps", g(full, "exchange_alt", "value"))
    return L


def compact_line(full):
    """The printed line: contract scalars, config, roofline (+ flat legs), cpu_baseline.  Never raises on a missing leg."""
    line = {k: (num(full.get(k), 6) if isinstance(full.get(k), float) else full.get(k)) for k in CONTRACT_KEYS if k in full}
    for k in ("metric", "unit", "dtype", "data", "scaling"):
        if k in line:
            line[k] = text(line[k], 110)
    cfg = full.get("config") or {}
    ex = cfg.get("exchange") or {}
    config = {"workload": text(cfg.get("workload"), 200), "rows_total": cfg.get("rows_total"), "rows_per_gpu": cfg.get("rows_per_gpu"),
              "queries_per_step": cfg.get("queries_per_step"), "k": cfg.get("k"), "parallelism": text(cfg.get("parallelism"), 40)}
    if ex:
        config["exchange"] = {k: v for k, v in (("kind", text(ex.get("kind"), 120)), ("rccl_ranks", ex.get("rccl_ranks", 0 if "ranks" in ex else None)), ("ranks", ex.get("ranks")),
                                                ("bytes_per_rank_per_step", ex.get("bytes_per_rank_per_step")),
                                                ("rccl_unavailable", text(ex.get("rccl_unavailable"), 120))) if v is not None}
    line["config"] = config
    if "verified_vs_exact_kernel" in full:
        line["verified_vs_exact_kernel"] = full["verified_vs_exact_kernel"]
    rf = full.get("roofline") or {}
    roof = {"bound": rf.get("bound"), "kernel": text(rf.get("kernel"), 60), "achieved": num(rf.get("achieved"), 5), "peak": num(rf.get("peak"), 5),
            "unit": rf.get("unit"), "frac": num(rf.get("frac")), "traffic": num(rf.get("traffic"), 5),
            "bytes_per_launch": rf.get("bytes_per_launch"), "avg_launch_ms": num(rf.get("avg_launch_ms"), 5), "launches_timed": rf.get("launches_timed"),
            "mfma_tflops": num(rf.get("mfma_tflops")), "mfma_frac": num(rf.get("mfma_frac")), "legs": flat_legs(full)}
    line["roofline"] = roof
    cb = full.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {"value": num(cb.get("value"), 5), "unit": text(cb.get("unit"), 100), "cores": cb.get("cores"), "kind": cb.get("kind"),
                                "sample": text(cb.get("sample"), 200)}
    if full.get("note"):
        line["note"] = text(full["note"], 120)
    line["detail"] = text(full.get("detail_file"), 60)
    # trim: legs go from the end (least important last) until the line is inside the target; then the free-text fields
    legs = roof["legs"]
    while len(json.dumps(line)) > LINE_TARGET and legs:
        legs.pop(next(reversed(legs)))
    if len(json.dumps(line)) > LINE_TARGET:
        for k in ("note", "detail"):
            line.pop(k, None)
        config.pop("exchange", None)
    s = json.dumps(line)
    assert len(s) <= LINE_LIMIT, len(s)
    return line
