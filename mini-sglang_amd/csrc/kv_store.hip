// KV slot write: k_pool[loc[t]] = k_src[t], v_pool[loc[t]] = v_src[t]  (reference memory/kvcache.py:69-79).
// HBM-bound byte copy: one 16-byte chunk per thread, K and V in the same launch.  Algorithmic bytes
// per token: 2 * Hkv*D*esize read + the same written + 8 (the slot index).
#include "common.h"

namespace m355 {

// row_bytes = Hkv*D*esize, a multiple of 16.  chunks = row_bytes / 16.
__global__ __launch_bounds__(256) void kv_store_kernel(uint8_t* __restrict__ k_pool, uint8_t* __restrict__ v_pool,
                                                       const uint8_t* __restrict__ k_src, const uint8_t* __restrict__ v_src,
                                                       int64_t k_row_bytes_stride, int64_t v_row_bytes_stride,
                                                       const int64_t* __restrict__ loc, int32_t T, int32_t chunks,
                                                       int64_t row_bytes, int64_t pool_row_bytes, int64_t S) {
    const int64_t total = (int64_t)T * chunks;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int32_t t = (int32_t)(i / chunks);
        const int32_t c = (int32_t)(i - (int64_t)t * chunks);
        int64_t slot = loc[t];
        if (slot < 0) slot += S;                       // torch index semantics
        if (slot < 0 || slot >= S) continue;           // never write outside the pool
        const u32x4 kk = *reinterpret_cast<const u32x4*>(k_src + t * k_row_bytes_stride + (int64_t)c * 16);
        const u32x4 vv = *reinterpret_cast<const u32x4*>(v_src + t * v_row_bytes_stride + (int64_t)c * 16);
        *reinterpret_cast<u32x4*>(k_pool + slot * pool_row_bytes + (int64_t)c * 16) = kk;
        *reinterpret_cast<u32x4*>(v_pool + slot * pool_row_bytes + (int64_t)c * 16) = vv;
    }
}

}  // namespace m355

extern "C" int m355_kv_store(void* k_pool, void* v_pool, int64_t pool_row_stride, const void* k_src, const void* v_src, int64_t k_row_stride,
                             int64_t v_row_stride, const int64_t* loc, int32_t T, int32_t Hkv, int32_t D, int64_t S,
                             int32_t dtype, void* stream) {
    if (T == 0) return M355_OK;
    if (!k_pool || !v_pool || !k_src || !v_src || !loc || T < 0 || Hkv <= 0 || D <= 0 || S <= 0) return M355_ERR_INVALID_ARG;
    int esize;
    switch (dtype) {
        case M355_DTYPE_BF16: case M355_DTYPE_F16: esize = 2; break;
        case M355_DTYPE_F32: esize = 4; break;
        default: return M355_ERR_UNSUPPORTED;
    }
    const int64_t row_bytes = (int64_t)Hkv * D * esize;
    if (row_bytes % 16 != 0) return M355_ERR_UNSUPPORTED;
    if (pool_row_stride == 0) pool_row_stride = (int64_t)Hkv * D;
    if (pool_row_stride < (int64_t)Hkv * D || (pool_row_stride * esize) % 16) return M355_ERR_INVALID_ARG;
    if (k_row_stride < (int64_t)Hkv * D || v_row_stride < (int64_t)Hkv * D) return M355_ERR_INVALID_ARG;
    if ((k_row_stride * esize) % 16 || (v_row_stride * esize) % 16) return M355_ERR_INVALID_ARG;
    if (((uintptr_t)k_pool | (uintptr_t)v_pool | (uintptr_t)k_src | (uintptr_t)v_src) & 15) return M355_ERR_INVALID_ARG;
    const int32_t chunks = (int32_t)(row_bytes / 16);
    const int64_t total = (int64_t)T * chunks;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(m355::kv_store_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (uint8_t*)k_pool,
                       (uint8_t*)v_pool, (const uint8_t*)k_src, (const uint8_t*)v_src, k_row_stride * esize,
                       v_row_stride * esize, loc, T, chunks, row_bytes, pool_row_stride * esize, S);
    return m355::check_launch();
}
