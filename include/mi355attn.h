/*
 * mi355attn.h - C ABI of libmi355attn.so: the MI355X (gfx950) attention hot path of Mini-SGLang.
 *
 * The reference (CrazyDave999/Mini-SGLang) has no native code and no FFI: this path is reached
 * through a Python operator interface (an attention-backend object).  Each entry point below names
 * the reference routine it replaces (paths relative to the reference checkout).  The Python class
 * that a maintainer plugs in at engine/model_runner.py:90-91 is mi355_attn.MI355AttnBackend, which
 * binds these symbols with ctypes (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its comment says "host";
 *   - strides are in ELEMENTS of the pointed-to type;
 *   - pool_row_stride: elements between consecutive token rows of a K or V pool ([S, Hkv, D] with a row pitch; 0 means
 *     contiguous, Hkv*D - the reference's own tensors, memory/kvcache.py:24-35).  A multiple of 8, >= Hkv*D.  A pool
 *     whose pitch is an ODD number of 128-byte lines (mi355_attn.KVCache pads 128 bytes per row) spreads every kv
 *     head's column over all address-interleave patterns; on MI355X a CU that streams ONE 128-byte column of a
 *     contiguous pool runs up to 30 % slower on some columns;
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*; NULL = the null stream);
 *     no entry point synchronises, allocates or frees, none reads the environment, and the library keeps no mutable
 *     state (only per-device look-ups of immutable attributes: CU count, kernel attributes, the scratch address);
 *   - return value: M355_OK (0) or a negative M355_ERR_* code; nothing throws or exits;
 *   - index dtypes are the reference's: page_table int32 (memory/page_manager.py:22-24);
 *     page_table_ids / seq_lens / prefix_lens / positions / out_cache_loc int64
 *     (engine/batch.py:258-273, 299-317).
 */
#ifndef MI355ATTN_H
#define MI355ATTN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define M355_VERSION 310 /* 0.3.1: the decode plan / workspace calls take the pools' row pitch (kernel choice follows the pool layout) */

enum {
    M355_OK = 0,
    M355_ERR_INVALID_ARG = -1, /* null pointer, negative size, misaligned pointer/stride        */
    M355_ERR_UNSUPPORTED = -2, /* head_dim / group size / dtype this build has no kernel for    */
    M355_ERR_WORKSPACE = -3,   /* workspace missing or smaller than m355_decode_workspace_bytes */
    M355_ERR_LAUNCH = -4,      /* hipGetLastError() after the launch was not hipSuccess         */
    M355_ERR_SCRATCH_LIMIT = -5 /* the call needs more than M355_STATIC_SCRATCH_BYTES of scratch */
};

/* Element type of q / K / V / out.  Attention: bf16 everywhere; IEEE half (Llama-2-style fp16 checkpoints, the dtype
 * the reference takes from the checkpoint, utils/model_config.py:40-42) through the default matrix-core kernels
 * (every attention entry point and m355_rope_kv_store).  f32 is accepted by m355_kv_store alone. */
enum { M355_DTYPE_BF16 = 0, M355_DTYPE_F16 = 1, M355_DTYPE_F32 = 2 };

/* out_cache_loc of a token that has no slot (its position lies past the page table's width, see m355_prefill_prepare).
 * The store kernels wrap negative slots the way torch indexing does (slot + S) and skip whatever is still outside
 * [0, S): INT64_MIN stays negative after the wrap for every S, so a token marked this way is never written anywhere
 * (-1 would wrap to the pool's LAST slot). */
#define M355_NO_SLOT INT64_MIN

int m355_version(void);
const char *m355_strerror(int code);

/* KV slot write.  Replaces KVCache.write_kv (memory/kvcache.py:69-79):
 *     k_pool[loc[t]] = k_src[t];  v_pool[loc[t]] = v_src[t]          for t in [0, T)
 * Pools are [S, Hkv, D] contiguous.  k_src / v_src are [T, Hkv*D] with a row stride (they are
 * views of the packed qkv projection, models/llama.py:133-134), inner stride 1, 16-byte aligned
 * rows.  Pure byte copy, bit-exact; negative slots wrap like torch indexing, out-of-range slots
 * (M355_NO_SLOT among them) are skipped.  dtype selects the element size only (bf16/f16: 2 bytes, f32: 4 bytes). */
int m355_kv_store(void *k_pool, void *v_pool, int64_t pool_row_stride, const void *k_src, const void *v_src,
                  int64_t k_row_stride, int64_t v_row_stride, const int64_t *loc, int32_t T,
                  int32_t Hkv, int32_t D, int64_t S, int32_t dtype, void *stream);

/* Slot index math.  Replaces the indexing expression at engine/batch.py:315-317 and :341-343:
 *     out[i] = (int64) page_table[row_of_token[i], positions[i] / page_size] * page_size
 *              + positions[i] % page_size                              for i in [0, n) */
int m355_out_cache_loc(const int32_t *page_table, int64_t pt_row_stride, const int64_t *row_of_token,
                       const int64_t *positions, int32_t page_size, int64_t n, int64_t *out,
                       void *stream);

/* Prefill metadata in one launch.  Replaces the per-request loop of Batch.prepare_for_prefill
 * (engine/batch.py:280-317) once the host has chosen rows and pages:
 *   plan (int32, device) = for each request r in [0, B): {row, prefix_len, seq_len, page_off}
 *   pages (int32, device) = concatenation over requests of cat(prefix_ppns, new_pages); request r
 *                           owns pages[page_off_r : page_off_r + ceil(seq_len_r / page_size)]
 *   cu_q (int32, device, [B+1]) = exclusive prefix sum of (seq_len - prefix_len)
 * Effects: page_table[row_r, 0:npages_r] = its pages (PageManager.write_ppns_prefill,
 * memory/page_manager.py:45-46); positions[cu_q[r] + j] = prefix_len_r + j;
 * row_of_token[...] = row_r; out_cache_loc[...] as above.  All outputs int64.
 * pt_cols = columns of the table (PageManager.max_page_num, memory/page_manager.py:21; <= pt_row_stride): entries
 * at or past it are not written and the token's out_cache_loc is M355_NO_SLOT (which m355_kv_store and
 * m355_rope_kv_store skip), so a request longer than the table is wide can never spill into the next request's row (the reference raises on the shape mismatch at page_manager.py:46; the
 * host planner here raises first). */
int m355_prefill_prepare(int32_t *page_table, int64_t pt_row_stride, int32_t pt_cols, const int32_t *plan,
                         const int32_t *pages, const int32_t *cu_q, int32_t B, int32_t page_size,
                         int64_t *positions, int64_t *row_of_token, int64_t *out_cache_loc,
                         void *stream);

/* Decode metadata in one launch.  Replaces Batch.prepare_for_decode (engine/batch.py:319-343):
 *     seq_lens_out[b] = seq_lens_in[b] + 1 (the two may alias); positions[b] = seq_lens_out[b] - 1;
 *     if new_pages[b] >= 0: page_table[rows[b], positions[b] / page_size] = new_pages[b]
 *     out_cache_loc[b] = page_table[rows[b], positions[b] / page_size] * page_size + positions[b] % page_size
 * new_pages[b] is the page the host allocator popped for request b, or -1 when the new token
 * still fits its last page (memory/kvcache.py:49-56).  pt_cols: as m355_prefill_prepare (a position past the table's
 * width writes nothing and gets out_cache_loc M355_NO_SLOT; the reference raises an index error at page_manager.py:54). */
int m355_decode_prepare(int32_t *page_table, int64_t pt_row_stride, int32_t pt_cols, const int64_t *rows,
                        const int64_t *seq_lens_in, int64_t *seq_lens_out, const int32_t *new_pages,
                        int32_t B, int32_t page_size, int64_t *positions, int64_t *out_cache_loc,
                        void *stream);

/* The free-page FIFO mirrored on the device, so that a decode step needs no page ids from the
 * host.  ring[capacity] holds page ids, state = {head, count} (int64).  Same order as the reference's list
 * (memory/kvcache.py:38-67, 87-91): init = range(page_num) (clear, :87-88), pop = take n from the front
 * (_allocate_pages, :58-65; the host keeps the ids, the device only advances), push = append at the back
 * (_free_pages, :66-67).  All calls are ordered on `stream`. */
int m355_page_ring_init(int32_t *ring, int64_t *state, int32_t capacity, void *stream);
int m355_page_ring_pop(int64_t *state, int32_t capacity, int32_t n, void *stream);
int m355_page_ring_push(int32_t *ring, int64_t *state, int32_t capacity, const int32_t *pages, int32_t n,
                        void *stream);

/* m355_decode_prepare with the allocator folded in: replaces Batch.prepare_for_decode (engine/batch.py:319-343)
 * INCLUDING allocate_pages_decode (memory/kvcache.py:49-56).  Walking the batch in order, request b takes the next
 * page off the ring iff seq_lens_in[b] % page_size == 0 (its new token starts a page); outputs as
 * m355_decode_prepare; state advances by the number of pages taken.  The caller guarantees that the ring holds
 * enough pages (the host knows count and lengths). */
int m355_decode_prepare_ring(int32_t *page_table, int64_t pt_row_stride, int32_t pt_cols, const int64_t *rows,
                             const int64_t *seq_lens_in, int64_t *seq_lens_out, const int32_t *ring,
                             int64_t *state, int32_t capacity, int32_t B, int32_t page_size,
                             int64_t *positions, int64_t *out_cache_loc, void *stream);

/* Decode attention over the paged KV pools.  Replaces TorchNativeAttnBackend.forward_decode /
 * _run_sdpa_forward_decode (layers/attention_backends/torch_native_backend.py:121-193, 237-276)
 * minus the KV write (call m355_kv_store first on the same stream):
 *     out[b, h, :] = softmax(q[b, h, :] . K_b[:, h / (Hq/Hkv), :]^T * scale) V_b[:, h / (Hq/Hkv), :]
 * over ALL seq_lens[b] cached tokens of request b, token p living in slot
 * page_table[page_table_ids[b], p / page_size] * page_size + p % page_size.
 * q / out: [B, Hq*D] with row strides; pools [S, Hkv, D].  The reference always uses
 * scale = D^-0.5 (SDPA default); it is an argument here so the caller states it.
 * Self-contained form, for callers without a per-forward hook: the call lays out the decode plan (m355_decode_plan, for
 * B * max_seq_len_hint tokens) inside `workspace` and then runs m355_decode_attn_planned over it - two small launches
 * more per call than the planned form, which shares one plan between all layers of a forward.
 * max_seq_len_hint must be >= max(seq_lens); workspace: m355_decode_workspace_bytes (same pool_row_stride as the call),
 * 256-byte aligned. */
int64_t m355_decode_workspace_bytes(int32_t B, int32_t Hq, int32_t Hkv, int32_t D, int32_t max_seq_len_hint,
                                    int64_t pool_row_stride);
int m355_decode_attn(const void *q, int64_t q_row_stride, const void *k_pool, const void *v_pool,
                     int64_t pool_row_stride, const int32_t *page_table, int64_t pt_row_stride, const int64_t *page_table_ids,
                     const int64_t *seq_lens, void *out, int64_t o_row_stride, int32_t B, int32_t Hq,
                     int32_t Hkv, int32_t D, int32_t page_size, float scale, int32_t max_seq_len_hint,
                     void *workspace, int64_t workspace_bytes, int32_t dtype, void *stream);

/* Planned decode attention: the same result as m355_decode_attn, with the batch's token stream laid out
 * once per forward (the page tables are shared by all layers) and then cut into equal ranges, one per wave, so
 * that the chip is filled exactly once with no per-chunk start-up.  This is the counterpart of the reference's
 * per-forward metadata step (FlashAttentionBackend.init_forward_metadata, flash_attention_backend.py:53-84).
 *   m355_decode_plan          once per forward, after the page table is final: builds `plan` (device memory of
 *                             m355_decode_plan_bytes bytes, 16-byte aligned) from seq_lens / page_table_ids /
 *                             page_table.  token_capacity must be >= sum(seq_lens) (a smaller value truncates).
 *                             pt_cols = columns of the table (as m355_prefill_prepare; <= pt_row_stride): a request
 *                             longer than pt_cols * page_size tokens is attended up to the row's end only - no look-up
 *                             ever leaves its row - and, like a request the capacity cut, it is left out of the fused
 *                             KV store (m355_decode_store_attn_planned writes nothing for it).
 *                             pool_row_stride = the row pitch (elements; 0 = contiguous [S, Hkv, D]) of the pools the
 *                             attention calls over this plan will read: which kernel serves the batch - and with it the
 *                             plan's format - also follows the pool layout (contiguous power-of-two rows of the
 *                             reference, memory/kvcache.py:24-35, pin a kv head to one address column; see
 *                             csrc/decode_stream.hip share_mode).
 *   m355_decode_attn_planned  once per layer, after m355_kv_store; same B, Hq, Hkv, D, token_capacity AND pool_row_stride
 *                             as the plan (a plan is only valid for pools of the pitch it was made for).
 *                             workspace: m355_decode_planned_workspace_bytes bytes, fp32 partials. */
int64_t m355_decode_plan_bytes(int32_t B, int32_t Hkv, int32_t D, int32_t G, int64_t token_capacity,
                               int64_t pool_row_stride);
int m355_decode_plan(const int64_t *seq_lens, const int64_t *page_table_ids, const int32_t *page_table,
                     int64_t pt_row_stride, int32_t pt_cols, int32_t B, int32_t page_size, int32_t Hkv, int32_t D,
                     int32_t G, int64_t token_capacity, int64_t pool_row_stride, void *plan, int64_t plan_bytes,
                     void *stream);
/* The plan's int32 word M355_PLAN_ERROR_WORD is 0 after m355_decode_plan.  The single-launch decode kernel hands
 * partial results of requests cut by a workgroup boundary from one workgroup to another; every wait in it is bounded,
 * and a wait that runs out (never seen on hardware; it would take another tenant holding CUs for seconds) sets this
 * word to 1 instead of hanging the GPU - the attention output of that launch is then not to be trusted.  Callers that
 * want to know copy the word back (4 bytes at plan + 4 * M355_PLAN_ERROR_WORD) whenever they next synchronise;
 * mi355_attn exposes it as DecodePlan.error() / MI355AttnBackend.check(). */
#define M355_PLAN_ERROR_WORD 6
int64_t m355_decode_planned_workspace_bytes(int32_t B, int32_t Hq, int32_t Hkv, int32_t D,
                                            int64_t token_capacity, int64_t pool_row_stride);
int m355_decode_attn_planned(const void *q, int64_t q_row_stride, const void *k_pool, const void *v_pool,
                             int64_t pool_row_stride, void *plan, void *out, int64_t o_row_stride, int32_t B, int32_t Hq,
                             int32_t Hkv, int32_t D, float scale, int64_t token_capacity, void *workspace,
                             int64_t workspace_bytes, int32_t dtype, void *stream);

/* KV slot write + planned decode attention in ONE launch: the same pool contents and the same output as
 * m355_kv_store(k_pool, v_pool, k_new, v_new, loc = the step's out_cache_loc) followed by m355_decode_attn_planned,
 * i.e. one whole TorchNativeAttnBackend.forward_decode (torch_native_backend.py:237-276).  The plan (built after
 * the step's page-table update) already knows every request's newest slot, so no loc argument is needed: the wave
 * that owns a request's last token reads that row from k_new / v_new [B, Hkv*D] (row strides in elements) instead
 * of the pool and writes it to the pool slot.  Saves the ~3.5 us store launch per layer.  Matrix-core kernel only
 * (M355_ERR_UNSUPPORTED under M355_STREAM_KERNEL=valu). */
int m355_decode_store_attn_planned(const void *q, int64_t q_row_stride, const void *k_new, int64_t k_row_stride,
                                   const void *v_new, int64_t v_row_stride, void *k_pool, void *v_pool,
                                   int64_t pool_row_stride, void *plan, void *out, int64_t o_row_stride, int32_t B, int32_t Hq,
                                   int32_t Hkv, int32_t D, float scale, int64_t token_capacity, void *workspace,
                                   int64_t workspace_bytes, int32_t dtype, void *stream);

/* m355_decode_store_attn_planned for callers that cannot allocate: code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc.  NO COUNTERPART IN THE REFERENCE (its backends allocate freely through
 * torch); same arguments minus the workspace, same pools and the same output bit for bit.  The fp32 partials live in a
 * device array inside the library instead - M355_STATIC_SCRATCH_BYTES per GPU the library has run on, held until the
 * process ends - and the call allocates and frees nothing.  A batch whose
 * m355_decode_planned_workspace_bytes(B, Hq, Hkv, D, token_capacity, pool_row_stride) exceeds that limit is refused with
 * M355_ERR_SCRATCH_LIMIT (nothing is launched); use the workspace form for it.  Every call on one GPU shares that
 * GPU's array: calls that overlap on one GPU (two streams, two threads) race - order them.  The first call on a
 * device looks the array's address up (hipGetSymbolAddress): make it outside graph capture. */
#define M355_STATIC_SCRATCH_BYTES (8 * 1024 * 1024)
int m355_decode_store_attn_planned_static(const void *q, int64_t q_row_stride, const void *k_new, int64_t k_row_stride,
                                          const void *v_new, int64_t v_row_stride, void *k_pool, void *v_pool,
                                          int64_t pool_row_stride, void *plan, void *out, int64_t o_row_stride, int32_t B, int32_t Hq,
                                          int32_t Hkv, int32_t D, float scale, int64_t token_capacity, int32_t dtype,
                                          void *stream);

/* Ragged prefill ("extend") attention over cached prefixes.  Replaces
 * TorchNativeAttnBackend.forward_extend / _run_sdpa_forward_extend
 * (torch_native_backend.py:25-119, 195-235) minus the KV write: packed query row cu_q[b] + j is
 * the token at absolute position prefix_lens[b] + j of request b and attends cached tokens
 * 0 .. prefix_lens[b] + j (its own K/V included - they are read back from the pool).
 * cu_q: int32 [B+1] exclusive prefix sum of (seq_lens - prefix_lens); T = cu_q[B];
 * max_extend_len >= max(seq_lens - prefix_lens) and max_seq_len_hint >= max(seq_lens) size the grid.
 * workspace: fp32 scratch of m355_prefill_workspace_bytes(...) bytes (may be NULL when that is 0; non-zero when the
 * launch is cut into KV splits - few query rows over long histories - whose partials a second kernel merges). */
int64_t m355_prefill_workspace_bytes(int32_t T, int32_t Hq, int32_t D, int32_t max_seq_len_hint);
int m355_prefill_attn(const void *q, int64_t q_row_stride, const void *k_pool, const void *v_pool,
                      int64_t pool_row_stride, const int32_t *page_table, int64_t pt_row_stride, const int64_t *page_table_ids,
                      const int64_t *seq_lens, const int64_t *prefix_lens, const int32_t *cu_q,
                      void *out, int64_t o_row_stride, int32_t B, int32_t T, int32_t max_extend_len,
                      int32_t max_seq_len_hint, int32_t Hq, int32_t Hkv, int32_t D, int32_t page_size,
                      float scale, void *workspace, int64_t workspace_bytes, int32_t dtype, void *stream);

/* Rotary embedding fused with the KV slot write (the step before the path).  Replaces
 * RotaryEmbedding.forward (layers/rotary_embedding.py:65-93; formula :8-18, neox style over the whole head:
 * y[i] = x[i]*cos - x[i+D/2]*sin, y[i+D/2] = x[i+D/2]*cos + x[i]*sin in fp32, cast back to bf16) as called at
 * models/llama.py:135, and - when k_pool is not NULL - also KVCache.write_kv (memory/kvcache.py:69-79) for the same
 * tokens, so the rotated k row is written to its slot without being re-read.
 *   q [T, Hq*D], k [T, Hkv*D]: rotated IN PLACE (row strides in elements, multiples of 8, 16-byte aligned);
 *   cos_sin_cache: fp32 [max_position, D], row p = cos(p*inv_freq[0..D/2)) | sin(...) (rotary_embedding.py:54-63);
 *   positions int64 [T] (clamped into [0, max_position));
 *   k_pool / v_pool [S, Hkv, D], v [T, Hkv*D], loc int64 [T]: as m355_kv_store; pass k_pool = NULL for rope only.
 * No FMA contraction: bit-identical to the formula evaluated in fp32 op by op.  Parity with the sgl_kernel CUDA
 * kernel the reference actually calls is unpinned (wheel not available). */
int m355_rope_kv_store(void *q, int64_t q_row_stride, void *k, int64_t k_row_stride, const void *v,
                       int64_t v_row_stride, const int64_t *positions, const float *cos_sin_cache,
                       int64_t max_position, void *k_pool, void *v_pool, int64_t pool_row_stride,
                       const int64_t *loc, int32_t T, int32_t Hq, int32_t Hkv, int32_t D, int64_t S, int32_t dtype, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MI355ATTN_H */
