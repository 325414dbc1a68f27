// Internal launch interface between the translation units of libmse_hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

namespace mse {

constexpr uint32_t ID_NONE = 0xFFFFFFFFu;
constexpr int TOPK_KMAX = 2048;    // largest k (incl. margin) one selection can return
constexpr int TOPK_FANOUT = 256;   // children per tournament group
constexpr int GROUP_ROWS = 32;     // base rows per group maximum written by the MFMA scan

// ---- scan_exact.hip ------------------------------------------------------------------------
int launch_scan_exact(const uint16_t* base, size_t n_rows, int d, const void* queries_dev, int nq, bool q_is_f32,
                      int64_t* scores, size_t score_stride, float* fscores, int n_cu, hipStream_t stream);
int launch_score_rows(const uint16_t* base, size_t n_rows, int d, const void* queries_dev, bool q_is_f32,
                      const uint32_t* ids_dev, size_t n_pairs, size_t pairs_per_query, int64_t* out, float* fout,
                      hipStream_t stream);

// ---- gen.hip ---------------------------------------------------------------------------------
int launch_generate_rows(uint16_t* out, uint32_t seed, uint64_t row0, size_t n_rows, int d, hipStream_t stream);
// *out_max_norm_bits = max over rows of ||row||_2 (float bits, atomicMax on non-negative floats); zero it first
int launch_row_norm_max(const uint16_t* base, size_t n_rows, int d, uint32_t* out_max_norm_bits, hipStream_t stream);
// eps[q] = factor * ||query_q||_2 * max_norm   (queries f16 [nq][d])
int launch_query_eps(const uint16_t* queries, int nq, int d, const uint32_t* max_norm_bits, float factor, float* eps,
                     hipStream_t stream);

// the same for f32 queries scanned as their f16 roundings q16: adds |q32 - q16| * max_norm (flat index)
int launch_query_eps_f32(const float* q32, const uint16_t* q16, int nq, int d, const uint32_t* max_norm_bits, float factor, float* eps,
                         hipStream_t stream);

// ---- topk.hip --------------------------------------------------------------------------------
enum KeyKind { KEY_I64 = 0, KEY_F32 = 1, KEY_U64 = 2, KEY_U32 = 3 };
// out[q][g] = max over in[q][g*F .. (g+1)*F) as order-preserving unsigned keys
// (u64 keys for KEY_I64/KEY_U64 input, u32 keys for KEY_F32/KEY_U32 input).
int launch_reduce_max(KeyKind kind, const void* in, size_t in_stride, size_t n_in, void* out, size_t out_stride,
                      size_t n_out, int nq, hipStream_t stream, size_t in_estride = 1);   // element (q, i) at in[q*in_stride + i*in_estride]
struct SelectArgs {
    KeyKind kind;            // type of `in` / `list_keys`
    const void* in;          // level array, [nq][in_stride]
    size_t in_stride, n_in;
    const uint32_t* parents; // [nq][par_stride] group ids (ID_NONE = empty), or nullptr
    size_t par_stride, n_par;
    int fanout;              // children per parent
    const uint32_t* list_ids;  // explicit candidates [nq][list_stride] (with list_keys), or nullptr
    const void* list_keys;
    size_t list_stride, n_list;
    size_t list_chunk = 0, list_chunk_stride = 0;  // if list_chunk != 0: candidate c lives at (c / chunk) * chunk_stride + c % chunk
    size_t list_id_chunk_stride = 0;               // chunk stride of list_ids when it differs from that of list_keys (0 = the same)
    int k;                   // number to select (<= TOPK_KMAX)
    uint32_t* out_ids;       // [nq][out_stride], best first, padded with ID_NONE
    void* out_keys;          // optional: raw keys (same type as `in`) of the selected, [nq][out_stride]
    size_t out_stride;
    int nq;
    // optional, per query, in the composite's score domain (sortable u64; 32-bit keys sit in the top half):
    // floor_hi: candidates whose score key is below it are skipped (the caller knows >= k candidates reach it -- e.g. the
    // k-th best key of the parent level, since every one of the k best parents has a child with exactly its key);
    // kth_hi_out: receives the score key of the k-th best candidate (0 when fewer than k candidates exist).
    const unsigned long long* floor_hi = nullptr;
    unsigned long long* kth_hi_out = nullptr;
};
int launch_select(const SelectArgs& a, hipStream_t stream);
// same, with element (q, i) of `in` at in[q*in_stride + i*in_estride] (group-major level arrays)
int launch_select_strided(const SelectArgs& a, size_t in_estride, hipStream_t stream);
// group-major float level [n_in][nq_pad] -> query-major u32 keys [nq][out_stride]
int launch_reduce_max_gq(const float* in, int nq_pad, size_t n_in, uint32_t* out, size_t out_stride, size_t n_out,
                         int nq, hipStream_t stream);
// ids[q][p*group + r] = parents[q][p]*group + r (ID_NONE if parent empty or row >= n_rows)
int launch_expand_groups(const uint32_t* parents, size_t par_stride, size_t n_par, int group, size_t n_rows,
                         uint32_t* ids, size_t ids_stride, int nq, hipStream_t stream);
// final packaging: out_scores[q][i] = keys (i64) ; out_ids[q][i] = ids + id_offset ; certificate margin
int launch_finalize(const uint32_t* sel_ids, const int64_t* sel_scores, size_t sel_stride, int k, int nq,
                    uint64_t id_offset, int64_t* out_scores, uint32_t* out_ids, size_t out_stride,
                    const float* group_keys, size_t gk_stride, int kg, size_t n_groups, const float* eps,
                    float* margin, hipStream_t stream);

// certificate margin for f32 keys: margin[q] = sel_keys[q][k-1] - (group_keys[q][kg-1] + eps[q])   (flat index)
int launch_margin_f32(const uint32_t* sel_ids, const float* sel_keys, size_t sel_stride, int k, int nq, const float* group_keys,
                      size_t gk_stride, int kg, size_t n_groups, const float* eps, float* margin, hipStream_t stream);

// per-query widening helpers: out[g][j] = in[g][idx[j]] (j < nb; -inf in the padding columns) | out row j = in row idx[j] (rows of
// row_bytes, a multiple of 16) | dst row idx[j] = src row j for the j with take[j] != 0 (take == nullptr: all)
int launch_gather_columns(const float* in, int nq_pad, size_t n_groups, const uint32_t* idx, int nb, float* out, int nbp, hipStream_t stream);
int launch_gather_rows16(const void* in, size_t row_bytes, const uint32_t* idx, int nb, void* out, hipStream_t stream);
int launch_scatter_topk(const uint32_t* idx, const uint8_t* take, int nb, int k, const int64_t* src_s, const uint32_t* src_i, int64_t* dst_s,
                        uint32_t* dst_i, size_t dst_stride, hipStream_t stream);

// a shard's [n] results on their way into a packed block: out_sc[i] = sc[i], out_ids[i] = ids[i] + id_offset; empty slots (ID_NONE, or
// sc == nullptr) become (INT64_MIN, ID_NONE)
int launch_block_finish(const int64_t* sc, const uint32_t* ids, size_t n, uint64_t id_offset, int64_t* out_sc, uint32_t* out_ids, hipStream_t stream);
int launch_scatter_rows4(const uint32_t* idx, const uint8_t* take, int nb, int k, const void* src, void* dst, size_t dst_stride, hipStream_t stream);

// ---- pq.hip ----------------------------------------------------------------------------------
int launch_pq_transform(const float* T, int d, const float* x, size_t n, float* out, hipStream_t stream);
int launch_pq_transform_vec(const float* T_transposed, int d, const float* x, float* out, hipStream_t stream);   // n = 1, same arithmetic
int launch_pq_lut(const float* centroids, int n_centroids, int d, int dpc, const float* t, float* lut,
                  hipStream_t stream);
int launch_pq_quantize(const float* centroids, int n_centroids, int d, int dpc, const float* t, size_t n,
                       uint8_t* codes, hipStream_t stream);
// nq > 1 (gathered ids only): query y uses table lut + y * n_chunks * n_centroids, ids + y * q_stride, out + y * q_stride
int launch_pq_adc(const float* lut, int n_chunks, int n_centroids, const uint8_t* codes, size_t n_codes,
                  const uint32_t* ids, size_t n, const uint8_t* desc, int n_desc, const float* scales, int64_t* out,
                  int n_cu, hipStream_t stream, int nq = 1, size_t q_stride = 0);
bool pq_scan_gmax_supported(int n_chunks, int n_centroids, const uint8_t* desc, int n_desc, const float* scales);
int launch_pq_scan_gmax(const float* lut, const uint8_t* codes, size_t n, const uint8_t* desc, const float* scales,
                        int64_t* gmax, int n_cu, hipStream_t stream);
int launch_pq_scan_gmax2(const float* lut0, const float* lut1, const uint8_t* codes, size_t n, const uint8_t* desc,
                         const float* scales, int64_t* gmax0, int64_t* gmax1, int n_cu, hipStream_t stream);
// four queries per pass: 12-bit integer nomination tables + certificate (pq.hip)
struct Pq4Params { double delta, c, eps; int ok; };
size_t pq4_table_bytes();
int launch_pq4_table(const float* luts, const float* scales, int n_valid, void* table, Pq4Params* params, hipStream_t stream, int nq = 4);
int launch_pq_scan_gmax4(const void* table, const uint8_t* codes, size_t n, const uint8_t* desc, uint32_t* gmax, int n_cu,
                         hipStream_t stream, int nq = 4);   // nq = 4: 12-bit tables, 8: 8-bit tables; gmax [n_groups][nq] (group-major)
int launch_pq4_certify(const Pq4Params* params, const uint32_t* group_keys, int n_nominated, int n_sel, const uint32_t* top_ids,
                       const int64_t* top_scores, size_t top_stride, int r, int nq, int* flag, hipStream_t stream);
int launch_add_descriptor(const uint32_t* ids, size_t n, const uint8_t* desc, int n_desc, size_t n_codes,
                          const float* scales, int64_t* out, hipStream_t stream);
int launch_f32_to_f16(const float* in, size_t n, uint16_t* out, hipStream_t stream);
int launch_f16_to_f32(const uint16_t* in, size_t n, float* out, hipStream_t stream);
int launch_pq_lut_batch(const float* centroids, int n_centroids, int d, int dpc, const float* t, size_t nq, float* lut,
                        hipStream_t stream);
int rank_max_targets();
int launch_rank(const int64_t* scores, size_t n, const uint32_t* targets, int m, unsigned long long* counts, int n_cu,
                hipStream_t stream);

// ---- disk_search.hip: the runtime de-duplication of the request path for a batch, on the device ------------------------------------
// per query q: similarity bits over its visited records (vis_ids [nq][cap], n_visited[q] of them, visit order), greedy keep-first
// filter; a dropped record becomes (ID_NONE, INT64_MIN) in place.  bits: dedup_batch_scratch_bytes(nq, cap) bytes of scratch.
size_t dedup_batch_scratch_bytes(size_t nq, size_t cap);
int launch_dedup_batch(const uint16_t* base, int d, uint32_t* vis_ids, long long* vis_scores, size_t cap, const uint32_t* n_visited, size_t nq,
                       float threshold, void* bits, hipStream_t st);
// ---- scan_mfma.hip ---------------------------------------------------------------------------
// group_max[q_pad_index][g] layout: [n_groups][nq_pad] floats (group-major), nq_pad multiple of 32
int launch_scan_mfma(const uint16_t* base, size_t n_rows, int d, const uint16_t* queries_dev, int nq_pad,
                     void* packed_scratch, float* group_max, int n_cu, hipStream_t stream,
                     hipEvent_t ev_begin = nullptr, hipEvent_t ev_end = nullptr, int gm_stride = 0 /* row stride of group_max; 0 = nq_pad */,
                     int n_pass = 1 /* passes in this launch: consecutive nq_pad-row query tiles, consecutive column ranges of group_max */);  // events bracket the scan kernel only
size_t mfma_packed_bytes(int d);
int mfma_query_tile(int d);  // most queries one pass handles at width d (320 or 256)
int mfma_pad(int nq, int d);   // padded query count of a pass of nq <= 256 queries: 128, 192 or 256

}  // namespace mse
