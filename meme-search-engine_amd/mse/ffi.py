"""ctypes binding of libmse_hip.so (include/mse.h).

This is the only way the Python host layer reaches the device: every compute entry point
below calls the C ABI.  There is NO CPU fallback -- if the shared library is missing or a call
fails, an exception is raised (MseError).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MSE_HIP_LIB: developer override (A/B builds of the same library); the default is the in-tree build
LIB_PATH = os.environ.get("MSE_HIP_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libmse_hip.so")


class MseError(RuntimeError):
    pass


_lib = None


class BuildConfig(C.Structure):
    """mse_build_config == IndexBuildConfig (diskann/src/lib.rs:42-52)."""
    _fields_ = [("r", C.c_uint64), ("l", C.c_uint64), ("maxc", C.c_uint64), ("alpha", C.c_int64), ("query_alpha", C.c_int64),
                ("saturate_graph", C.c_uint32), ("query_breakpoint", C.c_uint32), ("max_add_per_stitch_iter", C.c_uint64)]


u8p, u16p, u32p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint16), C.POINTER(C.c_uint32)
f32p, i64p = C.POINTER(C.c_float), C.POINTER(C.c_int64)
sz, vp = C.c_size_t, C.c_void_p

# name -> (restype, argtypes); kept in one table so tests can check it against include/mse.h
SIGNATURES = {
    "mse_last_error": (C.c_char_p, []),
    "mse_device_count": (C.c_int, []),
    "mse_set_device": (C.c_int, [C.c_int]),
    "mse_device_synchronize": (C.c_int, []),
    "mse_device_mem_info": (C.c_int, [C.POINTER(sz), C.POINTER(sz)]),
    "mse_version": (C.c_char_p, []),
    "mse_queries_per_pass_max": (sz, [sz]),
    "mse_scale_dot_f32": (C.c_int64, [C.c_float]),
    "mse_scale_dot_f64": (C.c_int64, [C.c_double]),
    "mse_base_from_host": (vp, [u16p, sz, sz]),
    "mse_base_wrap_device": (vp, [vp, sz, sz]),
    "mse_base_generate": (vp, [C.c_uint32, C.c_uint64, sz, sz]),
    "mse_base_free": (None, [vp]),
    "mse_base_len": (sz, [vp]),
    "mse_base_dim": (sz, [vp]),
    "mse_base_device_ptr": (vp, [vp]),
    "mse_base_read_rows": (C.c_int, [vp, sz, sz, u16p]),
    "mse_fast_dot_f16": (C.c_int, [u16p, u16p, sz, i64p]),
    "mse_searcher_new": (vp, [vp]),
    "mse_searcher_free": (None, [vp]),
    "mse_searcher_set_stream": (C.c_int, [vp, vp]),
    "mse_searcher_stream": (vp, [vp]),
    "mse_bruteforce_topk_f16": (C.c_int, [vp, u16p, sz, sz, C.c_int, i64p, u32p]),
    "mse_bruteforce_topk_f16_dev": (C.c_int, [vp, vp, sz, sz, C.c_int, C.c_uint64, vp, vp]),
    "mse_bruteforce_scores_f16": (C.c_int, [vp, u16p, i64p]),
    "mse_bruteforce_ranks_f16": (C.c_int, [vp, u16p, u32p, sz, u32p]),
    "mse_score_rows_f16": (C.c_int, [vp, u32p, sz, u16p, i64p]),
    "mse_merge_topk_dev": (C.c_int, [vp, vp, vp, sz, sz, sz, vp, vp]),
    "mse_topk_block_bytes": (sz, [sz, sz]),
    "mse_merge_topk_packed_dev": (C.c_int, [vp, vp, sz, sz, sz, vp, vp]),
    "mse_base_rows_changed": (C.c_int, [vp]),
    "mse_shard_group_new": (vp, [C.POINTER(C.c_int), sz, sz]),
    "mse_shard_group_free": (None, [vp]),
    "mse_shard_group_n_shards": (sz, [vp]),
    "mse_shard_group_len": (sz, [vp]),
    "mse_shard_group_device": (C.c_int, [vp, sz]),
    "mse_shard_group_peer_mapped": (C.c_int, [vp, sz]),
    "mse_shard_group_searcher": (vp, [vp, sz]),
    "mse_shard_group_generate": (C.c_int, [vp, C.c_uint32, C.c_uint64, sz]),
    "mse_shard_group_load_host": (C.c_int, [vp, u16p, sz]),
    "mse_shard_group_set_shard_device": (C.c_int, [vp, sz, vp, sz, C.c_uint64]),
    "mse_shard_group_search": (C.c_int, [vp, u16p, sz, sz, C.c_int, i64p, u32p]),
    "mse_shard_group_set_exchange": (C.c_int, [vp, C.c_int]),
    "mse_shard_group_exchange": (C.c_int, [vp]),
    "mse_shard_group_rccl_ranks": (C.c_int, [vp]),
    "mse_shard_group_last_timing": (C.c_int, [vp, C.POINTER(C.c_double)]),
    "mse_shard_group_search_dev": (C.c_int, [vp, vp, sz, sz, C.c_int, vp, vp]),
    "mse_comm_unique_id": (C.c_int, [vp]),
    "mse_comm_init": (vp, [vp, C.c_int, C.c_int]),
    "mse_comm_free": (None, [vp]),
    "mse_comm_rank": (C.c_int, [vp]),
    "mse_comm_size": (C.c_int, [vp]),
    "mse_comm_search_dev": (C.c_int, [vp, vp, vp, sz, sz, C.c_int, C.c_uint64, vp, vp]),
    "mse_comm_last_timing": (C.c_int, [vp, C.POINTER(C.c_double)]),
    "mse_debug_mfma_group_max": (C.c_int, [vp, u16p, sz, f32p]),
    "mse_searcher_scan_timing": (C.c_int, [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    "mse_searcher_last_stats": (C.c_int, [vp, u32p, u32p]),
    "mse_index_new": (vp, [C.c_int]),
    "mse_index_free": (None, [vp]),
    "mse_index_add": (C.c_int, [vp, f32p, sz]),
    "mse_index_ntotal": (sz, [vp]),
    "mse_index_search": (C.c_int, [vp, f32p, sz, sz, f32p, i64p]),
    "mse_index_stats": (C.c_int, [vp, C.POINTER(C.c_uint64)]),
    "mse_dispatcher_new": (vp, [vp, sz, C.c_uint32]),
    "mse_dispatcher_free": (None, [vp]),
    "mse_dispatcher_topk_f16": (C.c_int, [vp, u16p, sz, sz, i64p, u32p]),
    "mse_dispatcher_stats": (C.c_int, [vp, C.POINTER(C.c_uint64)]),
    "mse_dispatcher_searcher": (vp, [vp]),
    "mse_debug_dispatcher_fail_shared": (C.c_int, [vp, C.c_uint32]),
    "mse_debug_coalescer_selftest": (C.c_int, [C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "mse_pq_load": (vp, [f32p, sz, f32p, sz, sz]),
    "mse_pq_free": (None, [vp]),
    "mse_pq_apply_transform": (C.c_int, [vp, f32p, sz, f32p]),
    "mse_pq_quantize_batch": (C.c_int, [vp, f32p, sz, u8p]),
    "mse_pq_preprocess_query": (C.c_int, [vp, f32p, f32p]),
    "mse_pq_adc": (C.c_int, [vp, f32p, u8p, sz, i64p]),
    "mse_codes_from_host": (vp, [u8p, sz, sz, u8p, sz]),
    "mse_codes_free": (None, [vp]),
    "mse_codes_quantize_base": (vp, [vp, vp, u8p, sz]),
    "mse_pq_scan_timing": (C.c_int, [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    "mse_pq_scan_sustained": (C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    "mse_codes_len": (sz, [vp]),
    "mse_pq_adc_gather": (C.c_int, [vp, vp, f32p, f32p, u32p, sz, i64p]),
    "mse_pq_scan_topk": (C.c_int, [vp, vp, vp, f32p, f32p, sz, sz, i64p, u32p]),
    "mse_pq_scan_topk_batch": (C.c_int, [vp, vp, vp, f32p, sz, f32p, sz, sz, i64p, u32p]),
    "mse_pq_last_uncertified": (C.c_uint32, [vp]),
    "mse_debug_pq_group_max": (C.c_int, [vp, vp, f32p, f32p, f32p, i64p, i64p]),
    "mse_debug_pq4_group_max": (C.c_int, [vp, vp, f32p, f32p, C.c_int, C.c_int, u32p, C.POINTER(C.c_double)]),
    "mse_descriptor_product": (C.c_int64, [f32p, sz, u8p, C.c_uint32]),
    "mse_nb_new": (vp, [sz]),
    "mse_nb_free": (None, [vp]),
    "mse_nb_clear": (None, [vp]),
    "mse_nb_len": (sz, [vp]),
    "mse_nb_cap": (sz, [vp]),
    "mse_nb_insert": (None, [vp, C.c_uint32, C.c_int64]),
    "mse_nb_next_unvisited": (C.c_int, [vp, u32p]),
    "mse_nb_ids": (u32p, [vp]),
    "mse_nb_scores": (i64p, [vp]),
    "mse_greedy_search": (C.c_int, [vp, u32p, u32p, sz, C.c_uint32, u16p, C.c_int, C.c_uint32, vp, C.POINTER(sz)]),
    "mse_disk_greedy_search": (C.c_int, [vp, vp, vp, u32p, u32p, sz, u8p, C.c_uint32, u16p, f32p, f32p, C.c_int, sz, vp,
                                         u32p, i64p, sz, C.POINTER(sz), C.POINTER(sz), C.POINTER(sz)]),
    "mse_graph_from_host": (vp, [u32p, u32p, sz, sz, u8p]),
    "mse_graph_free": (None, [vp]),
    "mse_disk_search_batch": (C.c_int, [vp, vp, vp, vp, u32p, u16p, f32p, f32p, sz, C.c_int, sz, sz, u32p, i64p, u32p, u32p, i64p, sz,
                                        u32p, u32p, u32p]),
    "mse_disk_search_batch_f32": (C.c_int, [vp, vp, vp, vp, u32p, f32p, f32p, sz, C.c_int, sz, sz, u32p, i64p, u32p, u32p, i64p, sz,
                                            u32p, u32p, u32p]),
    "mse_graph_set_entries": (C.c_int, [vp, vp, u32p, sz]),
    "mse_disk_query_topk_block": (C.c_int, [vp, vp, vp, vp, u32p, u16p, f32p, f32p, sz, C.c_int, sz, sz, sz, C.c_uint64, vp, u32p, u32p, u32p]),
    "mse_pq_scan_topk_block": (C.c_int, [vp, vp, vp, f32p, sz, f32p, sz, sz, C.c_uint64, vp]),
    "mse_shard_group_base": (vp, [vp, sz]),
    "mse_shard_group_first_row": (C.c_uint64, [vp, sz]),
    "mse_comm_exchange_dev": (C.c_int, [vp, vp, vp, sz, sz, sz, vp, vp]),
    "mse_comm_pq_scan_topk": (C.c_int, [vp, vp, vp, vp, f32p, f32p, sz, sz, sz, C.c_uint64, vp, vp]),
    "mse_comm_query_topk": (C.c_int, [vp, vp, vp, vp, vp, u16p, f32p, f32p, sz, C.c_int, sz, sz, sz, C.c_uint64, vp, vp]),
    "mse_shard_group_attach_pq": (C.c_int, [vp, sz, vp, vp]),
    "mse_shard_group_attach_graph": (C.c_int, [vp, sz, vp]),
    "mse_shard_group_pq_scan_topk": (C.c_int, [vp, f32p, f32p, sz, sz, sz, i64p, u32p]),
    "mse_shard_group_query_topk": (C.c_int, [vp, u16p, f32p, f32p, sz, C.c_int, sz, sz, sz, i64p, u32p]),
    "mse_graph_set_entry_centroids": (C.c_int, [vp, f32p, sz, u32p, sz]),
    "mse_disk_query_topk_f32": (C.c_int, [vp, vp, vp, vp, u32p, f32p, f32p, sz, C.c_int, sz, sz, sz, u32p, i64p, u32p, u32p, u32p]),
    "mse_disk_query_submit_f32": (C.c_int, [vp, vp, vp, vp, f32p, f32p, sz, C.c_int, sz, sz, sz, u32p, i64p, u32p, u32p, u32p, vp, vp, C.POINTER(vp)]),
    "mse_disk_query_submit_f32_nocopy": (C.c_int, [vp, vp, vp, vp, f32p, f32p, sz, C.c_int, sz, sz, sz, u32p, i64p, u32p, u32p, u32p, vp, vp, C.POINTER(vp)]),
    "mse_completion_queue_new": (vp, []),
    "mse_completion_queue_free": (None, [vp]),
    "mse_completion_queue_fd": (C.c_int, [vp]),
    "mse_completion_queue_wait": (C.c_long, [vp, C.POINTER(vp), sz, C.c_long]),
    "mse_graph_completions": (C.c_long, [vp, C.POINTER(vp), sz, C.c_long]),
    "mse_graph_completion_fd": (C.c_int, [vp]),
    "mse_ticket_status": (C.c_int, [vp]),
    "mse_ticket_error": (C.c_char_p, [vp]),
    "mse_ticket_user": (vp, [vp]),
    "mse_ticket_free": (None, [vp]),
    "mse_graph_set_dedup": (C.c_int, [vp, C.c_float]),
    "mse_graph_set_coalescer": (C.c_int, [vp, sz, C.c_uint32, C.c_int]),
    "mse_graph_coalescer_stats": (C.c_int, [vp, C.POINTER(C.c_uint64)]),
    "mse_searcher_wait_stream": (C.c_int, [vp, vp]),
    "mse_searcher_beam_timing": (C.c_int, [vp, C.c_int, C.POINTER(C.c_uint64)]),
    "mse_debug_coalescer_selftest_workers": (C.c_int, [C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "mse_debug_coalescer_selftest_async": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_int, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "mse_disk_query_topk": (C.c_int, [vp, vp, vp, vp, u32p, u16p, f32p, f32p, sz, C.c_int, sz, sz, sz, u32p, i64p, u32p, u32p, u32p]),
    "mse_graph_new": (vp, [sz, sz]),
    "mse_graph_to_host": (C.c_int, [vp, u32p, u32p]),
    "mse_graph_len": (sz, [vp]),
    "mse_graph_max_degree": (sz, [vp]),
    "mse_graph_random_fill": (C.c_int, [vp, C.c_uint32, sz]),
    "mse_build_graph": (C.c_int, [vp, vp, u32p, sz, sz, C.c_uint32, vp]),
    "mse_robust_stitch": (C.c_int, [vp, vp, u32p, vp]),
    "mse_robust_prune": (C.c_int, [vp, u32p, i64p, sz, C.c_uint32, vp, u32p, C.POINTER(sz)]),
    "mse_graph_search_batch": (C.c_int, [vp, vp, u32p, u16p, sz, sz, C.c_int, C.c_uint32, u32p, i64p, u32p, u32p]),
    "mse_dedup_visited": (C.c_int, [vp, u32p, sz, C.c_float, u8p]),
    "mse_select_shard": (C.c_int, [f32p, sz, sz, f32p, C.POINTER(sz)]),
    "mse_medioid": (C.c_int, [vp, u32p]),
    "mse_score_model_load": (vp, [f32p, f32p, f32p, sz, sz, sz]),
    "mse_score_model_free": (None, [vp]),
    "mse_score_model_output_channels": (sz, [vp]),
    "mse_score_model_score_batch": (C.c_int, [vp, f32p, sz, f32p]),
    "mse_descriptor_buckets": (C.c_int, [f32p, sz, sz, f32p, sz, u8p]),
    "mse_siglip_create": (vp, [vp]),
    "mse_siglip_destroy": (None, [vp]),
    "mse_siglip_n_weights": (C.c_int, [vp]),
    "mse_siglip_weight_name": (C.c_char_p, [vp, C.c_int]),
    "mse_siglip_set_weight": (C.c_int, [vp, C.c_char_p, f32p, C.POINTER(sz), C.c_int]),
    "mse_siglip_finalize": (C.c_int, [vp]),
    "mse_siglip_encode_image": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, f32p, u16p]),
    "mse_siglip_encode_rgb8": (C.c_int, [vp, u8p, C.c_int, C.c_int, f32p, u16p]),
    "mse_bmp24_info": (C.c_int, [C.c_char_p, sz, u32p, u32p, u32p, C.POINTER(C.c_int)]),
    "mse_siglip_encode_bmp": (C.c_int, [vp, C.POINTER(C.c_char_p), C.POINTER(sz), C.c_int, C.c_int, f32p, u16p]),
    "mse_siglip_output_device": (vp, [vp, C.c_int]),
    "mse_siglip_stream": (vp, [vp]),
    "mse_siglip_debug_residual": (C.c_int, [vp, f32p]),
    "mse_debug_gemm_ms": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, f32p]),
    "mse_debug_gemm_small": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, f32p, C.POINTER(C.c_uint64)]),
    "mse_siglip_text_create": (vp, [vp]),
    "mse_siglip_text_destroy": (None, [vp]),
    "mse_siglip_text_n_weights": (C.c_int, [vp]),
    "mse_siglip_text_weight_name": (C.c_char_p, [vp, C.c_int]),
    "mse_siglip_text_set_weight": (C.c_int, [vp, C.c_char_p, f32p, C.POINTER(sz), C.c_int]),
    "mse_siglip_text_finalize": (C.c_int, [vp]),
    "mse_siglip_text_encode": (C.c_int, [vp, i64p, C.c_int, C.c_int, f32p, u16p]),
    "mse_siglip_text_encode_dev": (C.c_int, [vp, i64p, C.c_int, C.c_int]),
    "mse_siglip_text_output_device": (vp, [vp, C.c_int]),
    "mse_siglip_text_stream": (vp, [vp]),
}


class SiglipTextConfig(C.Structure):
    """mse_siglip_text_config (include/mse.h)"""
    _fields_ = [("width", C.c_int), ("layers", C.c_int), ("heads", C.c_int), ("mlp_dim", C.c_int),
                ("context_length", C.c_int), ("vocab_size", C.c_int), ("eps", C.c_float), ("gelu_tanh", C.c_int),
                ("max_batch", C.c_int)]


class SiglipConfig(C.Structure):
    """mse_siglip_config (include/mse.h)"""
    _fields_ = [("img_size", C.c_int), ("patch_size", C.c_int), ("in_chans", C.c_int), ("emb_dim", C.c_int),
                ("depth", C.c_int), ("num_heads", C.c_int), ("mlp_dim", C.c_int), ("eps", C.c_float),
                ("gelu_tanh", C.c_int), ("max_batch", C.c_int)]


def lib():
    """Load libmse_hip.so; raises MseError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MseError(
                f"{LIB_PATH} is missing: build it with __graft_entry__.build() "
                "(make -C meme-search-engine_amd/csrc).  There is no CPU fallback.")
        try:
            L = C.CDLL(LIB_PATH)
        except OSError as e:  # pragma: no cover - depends on the machine
            raise MseError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def last_error():
    msg = lib().mse_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc, what="call"):
    if rc != 0:
        raise MseError(f"{what} failed: {last_error()}")


def check_ptr(p, what="call"):
    if not p:
        raise MseError(f"{what} failed: {last_error()}")
    return p
