// Per-batch page-table metadata (reference engine/batch.py:244-343, memory/page_manager.py:45-54).
// Exact integer work; each kernel is launch-latency bound (a few KB of traffic).
#include "common.h"

namespace m355 {

__global__ __launch_bounds__(256) void out_cache_loc_kernel(const int32_t* __restrict__ page_table, int64_t pt_stride,
                                                            const int64_t* __restrict__ row_of_token,
                                                            const int64_t* __restrict__ positions, int32_t ps, int64_t n,
                                                            int64_t* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pos = positions[i];
        const int64_t vpn = pos / ps;
        out[i] = (int64_t)page_table[row_of_token[i] * pt_stride + vpn] * ps + (pos - vpn * ps);
    }
}

// plan: B+1 records of 4 int32 {row, prefix_len, seq_len, page_off}; record B carries only page_off = total pages.
__device__ __forceinline__ int find_req_by(const int32_t* key, int stride, int B, int32_t x) {
    // largest r in [0, B) with key[r*stride] <= x   (key non-decreasing, key[0] == 0)
    int lo = 0, hi = B - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (key[mid * stride] <= x) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// A request longer than the table is wide (pt_cols columns) never writes past its row: the out-of-range entries are
// dropped and its out_cache_loc is M355_NO_SLOT (INT64_MIN: still negative after the store kernels' torch-style wrap of
// negative slots, so m355_kv_store and m355_rope_kv_store skip it; -1 would wrap to the pool's last slot).  The host planner refuses such a batch first
// (mi355_attn/batch.py; the reference fails on the shape mismatch at memory/page_manager.py:46).
__global__ __launch_bounds__(256) void prefill_prepare_kernel(int32_t* __restrict__ page_table, int64_t pt_stride, int32_t pt_cols,
                                                              const int32_t* __restrict__ plan,
                                                              const int32_t* __restrict__ pages,
                                                              const int32_t* __restrict__ cu_q, int32_t B, int32_t ps,
                                                              int64_t* __restrict__ positions,
                                                              int64_t* __restrict__ row_of_token,
                                                              int64_t* __restrict__ out_cache_loc) {
    const int32_t total_pages = plan[B * 4 + 3];
    const int32_t T = cu_q[B];
    const int64_t work = (int64_t)total_pages + T;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < work; i += (int64_t)gridDim.x * blockDim.x) {
        if (i < total_pages) {
            // page-table row write: row[0:npages] = cat(prefix_ppns, new_pages)   (page_manager.py:45-46)
            int r = find_req_by(plan + 3, 4, B, (int32_t)i);
            // requests that own zero pages share a page_off with their successor: take the LAST such r whose
            // range is non-empty, i.e. the largest r with page_off[r] <= i - that is what find_req_by returns.
            const int32_t j = (int32_t)i - plan[r * 4 + 3];
            if (j < pt_cols) page_table[(int64_t)plan[r * 4 + 0] * pt_stride + j] = pages[i];
        } else {
            const int32_t t = (int32_t)(i - total_pages);
            const int r = find_req_by(cu_q, 1, B, t);          // largest r with cu_q[r] <= t (skips empty requests)
            const int32_t row = plan[r * 4 + 0];
            const int64_t pos = (int64_t)plan[r * 4 + 1] + (t - cu_q[r]);
            const int64_t vpn = pos / ps;
            positions[t] = pos;
            row_of_token[t] = row;
            out_cache_loc[t] = vpn < pt_cols ? (int64_t)pages[plan[r * 4 + 3] + vpn] * ps + (pos - vpn * ps) : (int64_t)M355_NO_SLOT;
        }
    }
}

__global__ __launch_bounds__(256) void decode_prepare_kernel(int32_t* __restrict__ page_table, int64_t pt_stride, int32_t pt_cols,
                                                             const int64_t* __restrict__ rows, const int64_t* seq_lens_in,
                                                             int64_t* seq_lens_out, const int32_t* __restrict__ new_pages, int32_t B, int32_t ps,
                                                             int64_t* __restrict__ positions,
                                                             int64_t* __restrict__ out_cache_loc) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int64_t L = seq_lens_in[b] + 1;                 // batch.py:325
    const int64_t pos = L - 1;                            // batch.py:326
    const int64_t vpn = pos / ps;
    seq_lens_out[b] = L;
    positions[b] = pos;
    if (vpn >= pt_cols) { out_cache_loc[b] = M355_NO_SLOT; return; }     // past the table's width: nothing written (see above)
    int32_t* entry = page_table + rows[b] * pt_stride + vpn;
    int32_t page = new_pages[b];
    if (page >= 0) *entry = page;                         // batch.py:331-337
    else page = *entry;
    out_cache_loc[b] = (int64_t)page * ps + (pos - vpn * ps);   // batch.py:341-343
}

// ------------------------------------------------------------------------------------------------
// Free-page FIFO kept on the device as well: ring[cap] of page ids, state = {head, count}.  Same order
// as the host FIFO (memory/kvcache.py:38-67: pop from the front, give back at the back), so the pages a decode step
// hands out here are exactly the ones the host allocator would have handed out, request by request in batch order.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void page_ring_init_kernel(int32_t* __restrict__ ring, int64_t* __restrict__ state, int32_t cap) {
    for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < cap; i += gridDim.x * blockDim.x) ring[i] = i;   // kvcache.py:87-88
    if (blockIdx.x == 0 && threadIdx.x == 0) { state[0] = 0; state[1] = cap; }
}

__global__ void page_ring_pop_kernel(int64_t* __restrict__ state, int32_t cap, int32_t n) {
    state[0] = (state[0] + n) % cap;
    state[1] -= n;
}

__global__ __launch_bounds__(256) void page_ring_push_kernel(int32_t* __restrict__ ring, int64_t* __restrict__ state, int32_t cap,
                                                             const int32_t* __restrict__ pages, int32_t n) {
    const int64_t tail = (state[0] + state[1]) % cap;
    __syncthreads();                                      // everyone has read the state before thread 0 rewrites it
    for (int32_t i = threadIdx.x; i < n; i += blockDim.x) ring[(tail + i) % cap] = pages[i];   // kvcache.py:67
    if (threadIdx.x == 0) state[1] += n;
}

// One workgroup walks the batch in order: a request whose new token starts a page takes the next page off the ring
// (engine/batch.py:325-343 with memory/kvcache.py:49-56 folded in); everything else is decode_prepare_kernel.
__global__ __launch_bounds__(1024) void decode_prepare_ring_kernel(int32_t* __restrict__ page_table, int64_t pt_stride, int32_t pt_cols,
                                                                   const int64_t* __restrict__ rows, const int64_t* seq_lens_in,
                                                                   int64_t* seq_lens_out, const int32_t* __restrict__ ring,
                                                                   int64_t* __restrict__ state, int32_t cap, int32_t B, int32_t ps,
                                                                   int64_t* __restrict__ positions,
                                                                   int64_t* __restrict__ out_cache_loc) {
    __shared__ int32_t wave_sum[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t head = state[0];
    int32_t taken = 0;                                    // pages handed out to requests before this round of 1024
    for (int base = 0; base < B; base += 1024) {
        const int b = base + tid;
        const int64_t pos = b < B ? seq_lens_in[b] : 1;   // old length = position of the new token
        const bool need = b < B && pos % ps == 0;         // ceil((pos+1)/ps) - ceil(pos/ps) == 1
        const unsigned long long mask = __ballot(need);
        if (lane == 0) wave_sum[wv] = __popcll(mask);
        __syncthreads();
        int32_t before = taken, total = 0;
        for (int k = 0; k < 16; ++k) { if (k < wv) before += wave_sum[k]; total += wave_sum[k]; }
        before += __popcll(mask & ((1ull << lane) - 1));
        if (b < B) {
            const int64_t vpn = pos / ps;
            seq_lens_out[b] = pos + 1;
            positions[b] = pos;
            if (vpn < pt_cols) {
                int32_t* entry = page_table + rows[b] * pt_stride + vpn;
                int32_t page;
                if (need) { page = ring[(head + before) % cap]; *entry = page; }
                else page = *entry;
                out_cache_loc[b] = (int64_t)page * ps + (pos - vpn * ps);
            } else {
                out_cache_loc[b] = M355_NO_SLOT;          // past the table's width (the page is still taken: FIFO order holds)
            }
        }
        taken += total;
        __syncthreads();                                  // wave_sum is rewritten next round
    }
    if (tid == 0) { state[0] = (head + taken) % cap; state[1] -= taken; }
}

}  // namespace m355

extern "C" int m355_page_ring_init(int32_t* ring, int64_t* state, int32_t capacity, void* stream) {
    if (!ring || !state || capacity <= 0) return M355_ERR_INVALID_ARG;
    int blocks = (capacity + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(m355::page_ring_init_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, ring, state, capacity);
    return m355::check_launch();
}

extern "C" int m355_page_ring_pop(int64_t* state, int32_t capacity, int32_t n, void* stream) {
    if (n == 0) return M355_OK;
    if (!state || capacity <= 0 || n < 0 || n > capacity) return M355_ERR_INVALID_ARG;
    hipLaunchKernelGGL(m355::page_ring_pop_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state, capacity, n);
    return m355::check_launch();
}

extern "C" int m355_page_ring_push(int32_t* ring, int64_t* state, int32_t capacity, const int32_t* pages, int32_t n,
                                   void* stream) {
    if (n == 0) return M355_OK;
    if (!ring || !state || !pages || capacity <= 0 || n < 0 || n > capacity) return M355_ERR_INVALID_ARG;
    hipLaunchKernelGGL(m355::page_ring_push_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, ring, state, capacity, pages, n);
    return m355::check_launch();
}

extern "C" int m355_decode_prepare_ring(int32_t* page_table, int64_t pt_row_stride, int32_t pt_cols, const int64_t* rows,
                                        const int64_t* seq_lens_in, int64_t* seq_lens_out, const int32_t* ring,
                                        int64_t* state, int32_t capacity, int32_t B, int32_t page_size, int64_t* positions,
                                        int64_t* out_cache_loc, void* stream) {
    if (B == 0) return M355_OK;
    if (!page_table || !rows || !seq_lens_in || !seq_lens_out || !ring || !state || !positions || !out_cache_loc || B < 0 ||
        page_size <= 0 || pt_row_stride <= 0 || pt_cols <= 0 || pt_cols > pt_row_stride || capacity <= 0)
        return M355_ERR_INVALID_ARG;
    hipLaunchKernelGGL(m355::decode_prepare_ring_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, page_table, pt_row_stride,
                       pt_cols, rows, seq_lens_in, seq_lens_out, ring, state, capacity, B, page_size, positions, out_cache_loc);
    return m355::check_launch();
}

extern "C" int m355_out_cache_loc(const int32_t* page_table, int64_t pt_row_stride, const int64_t* row_of_token,
                                  const int64_t* positions, int32_t page_size, int64_t n, int64_t* out, void* stream) {
    if (n == 0) return M355_OK;
    if (!page_table || !row_of_token || !positions || !out || n < 0 || page_size <= 0 || pt_row_stride <= 0)
        return M355_ERR_INVALID_ARG;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(m355::out_cache_loc_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, page_table,
                       pt_row_stride, row_of_token, positions, page_size, n, out);
    return m355::check_launch();
}

extern "C" int m355_prefill_prepare(int32_t* page_table, int64_t pt_row_stride, int32_t pt_cols, const int32_t* plan, const int32_t* pages,
                                    const int32_t* cu_q, int32_t B, int32_t page_size, int64_t* positions,
                                    int64_t* row_of_token, int64_t* out_cache_loc, void* stream) {
    if (B == 0) return M355_OK;
    if (!page_table || !plan || !pages || !cu_q || !positions || !row_of_token || !out_cache_loc || B < 0 ||
        page_size <= 0 || pt_row_stride <= 0 || pt_cols <= 0 || pt_cols > pt_row_stride)
        return M355_ERR_INVALID_ARG;
    // the host does not pass T/total pages (they sit in the device plan); size the grid generously and grid-stride
    hipLaunchKernelGGL(m355::prefill_prepare_kernel, dim3(512), dim3(256), 0, (hipStream_t)stream, page_table,
                       pt_row_stride, pt_cols, plan, pages, cu_q, B, page_size, positions, row_of_token, out_cache_loc);
    return m355::check_launch();
}

extern "C" int m355_decode_prepare(int32_t* page_table, int64_t pt_row_stride, int32_t pt_cols, const int64_t* rows,
                                   const int64_t* seq_lens_in, int64_t* seq_lens_out, const int32_t* new_pages, int32_t B,
                                   int32_t page_size, int64_t* positions, int64_t* out_cache_loc, void* stream) {
    if (B == 0) return M355_OK;
    if (!page_table || !rows || !seq_lens_in || !seq_lens_out || !new_pages || !positions || !out_cache_loc || B < 0 || page_size <= 0 ||
        pt_row_stride <= 0 || pt_cols <= 0 || pt_cols > pt_row_stride)
        return M355_ERR_INVALID_ARG;
    hipLaunchKernelGGL(m355::decode_prepare_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, page_table,
                       pt_row_stride, pt_cols, rows, seq_lens_in, seq_lens_out, new_pages, B, page_size, positions, out_cache_loc);
    return m355::check_launch();
}
