// Declaration of the row-parallel paged attention launcher shared by decode_attn.hip and prefill_attn.hip.
#pragma once
#include "common.h"

namespace m355 {

constexpr int kRowsMaxChunk = 512;

int paged_rows_attn(const void* q, int64_t q_stride, const void* k_pool, const void* v_pool, int64_t pool_row_stride,
                    const int32_t* page_table,
                    int64_t pt_stride, const int64_t* row_ids, const int64_t* kv_lens, const int32_t* cu_q,
                    const int64_t* prefix_lens, int32_t B, void* out, int64_t o_stride, int64_t NQ, int32_t Hq,
                    int32_t Hkv, int32_t D, int32_t ps, float scale, int32_t max_len_hint, int32_t chunk,
                    void* workspace, int64_t workspace_bytes, int32_t dtype, hipStream_t st);

}  // namespace m355
