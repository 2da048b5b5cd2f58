// Neox-style rotary embedding on q and k fused with the KV slot write (the step right before the
// attention path; reference layers/rotary_embedding.py:8-18 formula, :65-93 call, models/llama.py:133-136).
//
//   for every token t, head h, i in [0, D/2):   c = cache[pos[t]][i],  s = cache[pos[t]][D/2 + i]      (fp32)
//       y[i]       = x[i] * c - x[i + D/2] * s
//       y[i + D/2] = x[i + D/2] * c + x[i] * s                                     (fp32, then cast to bf16)
//   q and k are rotated IN PLACE (as the reference's kernel does); when pools are given the rotated k row and the
//   v row are also written to slot loc[t] in the same launch, so k is not re-read by a separate kv_store.
//
// HBM-bound elementwise work: one thread = 8 elements of the first half of a head + their 8 partners (two 16-byte
// loads, two 16-byte stores, 64 bytes of cos/sin that hit L2 for all heads of a token).  Algorithmic bytes per
// token: 2*(Hq+Hkv)*D*2 (q, k in and out) + Hkv*D*2 (v) + 2*Hkv*D*2 (pool writes) + 4*D (cache row) + 16.
// No fused multiply-add contraction: results are bit-identical to the fp32 formula evaluated op by op.
#include "common.h"

namespace m355 {

#pragma clang fp contract(off)
template <bool F16>
__device__ __forceinline__ void rotate8(u32x4& lo, u32x4& hi, const float* __restrict__ c, const float* __restrict__ s) {
#pragma clang fp contract(off)
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const float a0 = unpack16_lo<F16>(lo[w]), a1 = unpack16_hi<F16>(lo[w]);
        const float b0 = unpack16_lo<F16>(hi[w]), b1 = unpack16_hi<F16>(hi[w]);
        const float c0 = c[2 * w], c1 = c[2 * w + 1], s0 = s[2 * w], s1 = s[2 * w + 1];
        const float y0 = a0 * c0 - b0 * s0, y1 = a1 * c1 - b1 * s1;
        const float z0 = b0 * c0 + a0 * s0, z1 = b1 * c1 + a1 * s1;
        lo[w] = pack16<F16>(y0, y1);
        hi[w] = pack16<F16>(z0, z1);
    }
}

// work items per token: (Hq + Hkv) * D/16 rotations, then Hkv * D/8 v-row chunks (only when pools are given)
template <bool STORE, bool F16 = false>
__global__ __launch_bounds__(256) void rope_kv_store_kernel(uint16_t* __restrict__ q, int64_t q_stride, uint16_t* __restrict__ k,
                                                            int64_t k_stride, const uint16_t* __restrict__ v, int64_t v_stride,
                                                            const int64_t* __restrict__ positions,
                                                            const float* __restrict__ cache, int64_t max_pos,
                                                            uint16_t* __restrict__ k_pool, uint16_t* __restrict__ v_pool,
                                                            const int64_t* __restrict__ loc, int32_t T, int32_t Hq, int32_t Hkv,
                                                            int32_t D, int64_t S, int64_t pool_stride) {
    const int32_t half8 = D / 16;                        // 8-element chunks in half a head
    const int32_t rot_items = (Hq + Hkv) * half8;
    const int32_t v_items = STORE ? Hkv * (D / 8) : 0;
    const int32_t per_tok = rot_items + v_items;
    const int64_t total = (int64_t)T * per_tok;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int32_t t = (int32_t)(i / per_tok);
        const int32_t j = (int32_t)(i - (int64_t)t * per_tok);
        int64_t slot = 0;
        bool slot_ok = false;
        if constexpr (STORE) {
            slot = loc[t];
            if (slot < 0) slot += S;
            slot_ok = slot >= 0 && slot < S;
        }
        if (j < rot_items) {
            const int32_t head = j / half8, c8 = j - head * half8;
            int64_t pos = positions[t];
            pos = pos < 0 ? 0 : (pos >= max_pos ? max_pos - 1 : pos);         // never read outside the cache
            const float* crow = cache + pos * D + c8 * 8;
            float cs[8], sn[8];
            *reinterpret_cast<f32x4*>(cs) = *reinterpret_cast<const f32x4*>(crow);
            *reinterpret_cast<f32x4*>(cs + 4) = *reinterpret_cast<const f32x4*>(crow + 4);
            *reinterpret_cast<f32x4*>(sn) = *reinterpret_cast<const f32x4*>(crow + D / 2);
            *reinterpret_cast<f32x4*>(sn + 4) = *reinterpret_cast<const f32x4*>(crow + D / 2 + 4);
            const bool is_q = head < Hq;
            uint16_t* row = is_q ? q + t * q_stride + (int64_t)head * D : k + t * k_stride + (int64_t)(head - Hq) * D;
            u32x4 lo = *reinterpret_cast<const u32x4*>(row + c8 * 8);
            u32x4 hi = *reinterpret_cast<const u32x4*>(row + D / 2 + c8 * 8);
            rotate8<F16>(lo, hi, cs, sn);
            *reinterpret_cast<u32x4*>(row + c8 * 8) = lo;
            *reinterpret_cast<u32x4*>(row + D / 2 + c8 * 8) = hi;
            if constexpr (STORE) {
                if (!is_q && slot_ok) {
                    uint16_t* dst = k_pool + slot * pool_stride + (int64_t)(head - Hq) * D;
                    *reinterpret_cast<u32x4*>(dst + c8 * 8) = lo;
                    *reinterpret_cast<u32x4*>(dst + D / 2 + c8 * 8) = hi;
                }
            }
        } else if constexpr (STORE) {
            const int32_t c = j - rot_items;                                    // 16-byte chunk of the v row
            if (slot_ok)
                *reinterpret_cast<u32x4*>(v_pool + slot * pool_stride + (int64_t)c * 8) =
                    *reinterpret_cast<const u32x4*>(v + t * v_stride + (int64_t)c * 8);
        }
    }
}

}  // namespace m355

extern "C" int m355_rope_kv_store(void* q, int64_t q_row_stride, void* k, int64_t k_row_stride, const void* v,
                                  int64_t v_row_stride, const int64_t* positions, const float* cos_sin_cache,
                                  int64_t max_position, void* k_pool, void* v_pool, int64_t pool_row_stride, const int64_t* loc,
                                  int32_t T, int32_t Hq, int32_t Hkv, int32_t D, int64_t S, int32_t dtype, void* stream) {
    if (T == 0) return M355_OK;
    if (!q || !k || !positions || !cos_sin_cache || T < 0 || Hq <= 0 || Hkv <= 0 || max_position <= 0) return M355_ERR_INVALID_ARG;
    const bool store = k_pool != nullptr;
    if (store && (!v_pool || !v || !loc || S <= 0)) return M355_ERR_INVALID_ARG;
    if (dtype != M355_DTYPE_BF16 && dtype != M355_DTYPE_F16) return M355_ERR_UNSUPPORTED;
    const bool f16 = dtype == M355_DTYPE_F16;
    if (D <= 0 || D % 16) return M355_ERR_UNSUPPORTED;
    if (q_row_stride < (int64_t)Hq * D || k_row_stride < (int64_t)Hkv * D || q_row_stride % 8 || k_row_stride % 8) return M355_ERR_INVALID_ARG;
    if (store && (v_row_stride < (int64_t)Hkv * D || v_row_stride % 8)) return M355_ERR_INVALID_ARG;
    if (pool_row_stride == 0) pool_row_stride = (int64_t)Hkv * D;
    if (store && (pool_row_stride < (int64_t)Hkv * D || pool_row_stride % 8)) return M355_ERR_INVALID_ARG;
    uintptr_t bits = (uintptr_t)q | (uintptr_t)k | (uintptr_t)cos_sin_cache;
    if (store) bits |= (uintptr_t)v | (uintptr_t)k_pool | (uintptr_t)v_pool;
    if (bits & 15) return M355_ERR_INVALID_ARG;
    const int64_t per_tok = (int64_t)(Hq + Hkv) * (D / 16) + (store ? (int64_t)Hkv * (D / 8) : 0);
    int64_t blocks = ((int64_t)T * per_tok + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (f16) {
        if (store)
            hipLaunchKernelGGL((m355::rope_kv_store_kernel<true, true>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                               (uint16_t*)q, q_row_stride, (uint16_t*)k, k_row_stride, (const uint16_t*)v, v_row_stride, positions,
                               cos_sin_cache, max_position, (uint16_t*)k_pool, (uint16_t*)v_pool, loc, T, Hq, Hkv, D, S, pool_row_stride);
        else
            hipLaunchKernelGGL((m355::rope_kv_store_kernel<false, true>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                               (uint16_t*)q, q_row_stride, (uint16_t*)k, k_row_stride, (const uint16_t*)nullptr, (int64_t)0, positions,
                               cos_sin_cache, max_position, (uint16_t*)nullptr, (uint16_t*)nullptr, (const int64_t*)nullptr, T, Hq, Hkv,
                               D, (int64_t)0, (int64_t)0);
        return m355::check_launch();
    }
    if (store)
        hipLaunchKernelGGL(m355::rope_kv_store_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                           (uint16_t*)q, q_row_stride, (uint16_t*)k, k_row_stride, (const uint16_t*)v, v_row_stride, positions,
                           cos_sin_cache, max_position, (uint16_t*)k_pool, (uint16_t*)v_pool, loc, T, Hq, Hkv, D, S, pool_row_stride);
    else
        hipLaunchKernelGGL(m355::rope_kv_store_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                           (uint16_t*)q, q_row_stride, (uint16_t*)k, k_row_stride, (const uint16_t*)nullptr, (int64_t)0, positions,
                           cos_sin_cache, max_position, (uint16_t*)nullptr, (uint16_t*)nullptr, (const int64_t*)nullptr, T, Hq, Hkv,
                           D, (int64_t)0, (int64_t)0);
    return m355::check_launch();
}
