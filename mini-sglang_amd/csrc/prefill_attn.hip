// Ragged prefill ("extend") attention over cached prefixes (reference torch_native_backend.py:25-119, 195-235).
#include "common.h"
#include "paged_rows.h"

extern "C" int64_t m355_prefill_workspace_bytes(int32_t T, int32_t Hq, int32_t D, int32_t max_seq_len_hint) {
    if (T <= 0 || Hq <= 0 || D <= 0 || max_seq_len_hint <= 0) return 0;
    const int64_t splits = (max_seq_len_hint + m355::kRowsMaxChunk - 1) / m355::kRowsMaxChunk;
    if (splits <= 1) return 0;
    return (int64_t)T * Hq * splits * (D + 2) * (int64_t)sizeof(float);
}

extern "C" int m355_prefill_attn(const void* q, int64_t q_row_stride, const void* k_pool, const void* v_pool,
                                 const int32_t* page_table, int64_t pt_row_stride, const int64_t* page_table_ids,
                                 const int64_t* seq_lens, const int64_t* prefix_lens, const int32_t* cu_q, void* out,
                                 int64_t o_row_stride, int32_t B, int32_t T, int32_t max_extend_len,
                                 int32_t max_seq_len_hint, int32_t Hq, int32_t Hkv, int32_t D, int32_t page_size,
                                 float scale, void* workspace, int64_t workspace_bytes, int32_t dtype, void* stream) {
    if (T == 0 || B == 0) return M355_OK;
    if (!seq_lens || !prefix_lens || !cu_q || !page_table_ids || B < 0 || T < 0 || max_extend_len <= 0)
        return M355_ERR_INVALID_ARG;
    // Row-parallel path: every packed query row is handled like a decode query of length prefix + j + 1.
    return m355::paged_rows_attn(q, q_row_stride, k_pool, v_pool, page_table, pt_row_stride, page_table_ids, nullptr,
                                 cu_q, prefix_lens, B, out, o_row_stride, T, Hq, Hkv, D, page_size, scale,
                                 max_seq_len_hint, m355::kRowsMaxChunk, workspace, workspace_bytes, dtype,
                                 (hipStream_t)stream);
}
