// o_proj + SUM all-reduce in ONE launch per rank (include/mi355tp.h): a skinny bf16 GEMM on the matrix
// cores whose epilogue pushes every 16 x 64 output tile to all ranks over xGMI peer pointers, then reduces the tile of
// every rank in rank order.  Replaces F.linear + dist.all_reduce (reference layers/linear.py:228-232) at decode sizes,
// where the collective is latency-bound: no second kernel, no ring steps, every link of the GPU busy at once.
//
//   grid  (N / 64, ceil(T / 16), ranks in this launch), 320 threads = 4 working waves + 1 that fetches the call's epoch
//   GEMM  the workgroup's 4 waves split K; each runs  acc[j] += W_j (16 x 32) . X^T (32 x 16)  with
//         v_mfma_f32_16x16x32 - W as the A operand, so a lane ends up with 4 CONSECUTIVE output columns of one token -
//         and the four partial tiles are added through LDS in a fixed order
//   push  128 threads hold 16 bytes (8 columns of one row) each; the tile, rounded once to the element type, goes to slot
//         `rank` of every rank's receive buffer in 128-byte row pieces; system fence; one flag per (source rank, tile)
//   wait  bounded polling of this rank's own flags for the tile; then the `world` slots are added in fp32, rank 0 first
//
// Weight traffic dominates (N x K elements per rank, read once); x is re-read by the N/64 workgroups of a row block from L2.
#include <string.h>

#include <map>
#include <mutex>
#include <unordered_map>
#include <utility>
#include <vector>

#include "common.h"
#include "../../include/mi355tp.h"

namespace m355 {

struct OprojParams {
    M355OprojRank r[M355_MAX_RANKS];
    int32_t world, T, N, K, T_cap;
    uint32_t epoch;
    uint32_t* epoch_state;                   // device [world][tiles_cap] calls made per (rank, tile), or null: `epoch` counts on the host
    int32_t* error;
};

constexpr int kOprojPollLimit = 1 << 22;     // ~1 s of sleeping polls, then the wait gives up and raises *error

__device__ __forceinline__ uint32_t ld_sys(const uint32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// 16-byte accesses at SYSTEM scope (sc0 sc1: never served from, never left in, this GPU's caches).  The receive buffers are
// uncached allocations on top of that.  No cache-wide write-back / invalidate is used anywhere in this kernel: a
// system-scope release or acquire FENCE flushes / drops the whole L2 on gfx950, once per workgroup - measured 33 us per call.
__device__ __forceinline__ void st16_sys(void* p, u32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ u32x4 ld16_sys(const void* p) {
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// (one workgroup per CU is all a launch has: tell the register allocator, or it packs the loop into 64 registers for an
// occupancy nobody uses and waits for every load before the next is issued)
template <bool F16>
__global__ __launch_bounds__(320) __attribute__((amdgpu_waves_per_eu(1, 4))) void oproj_allreduce_kernel(const OprojParams p) {
    __shared__ __attribute__((aligned(16))) float red[4][16][68];
    __shared__ uint32_t epoch_s;
    __shared__ int gave_up_s;
    if (threadIdx.x == 0) gave_up_s = 0;                 // (published by the barrier behind the GEMM)
    const M355OprojRank& R = p.r[blockIdx.z];
    // the call's epoch: the host's count, or - so that ONE captured launch can be replayed - a count per (rank, tile) in
    // device memory that the tile's own workgroup reads and advances (wave 4, below): no other workgroup of any launch
    // on this device touches the word, every rank makes the same calls, so the counts agree across ranks tile by tile
    const int tiles_x = p.N / 64;
    const int tile = blockIdx.y * tiles_x + blockIdx.x;
    const int tiles_cap = ((p.T_cap + 15) / 16) * tiles_x;
    uint32_t* const own_epoch = p.epoch_state ? p.epoch_state + (int64_t)R.rank * tiles_cap + tile : nullptr;
    uint32_t epoch = p.epoch;                            // (the device count: wave 4 below)
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int col0 = blockIdx.x * 64, row0 = blockIdx.y * 16;
    const int n = lane & 15, kg = lane >> 4;
    // ---- wave 4 fetches the call's count while waves 0-3 multiply.  Agent-scope accesses: the word was written by a
    // workgroup of the previous launch, possibly on another XCD - a round trip to memory that a GEMM wave would have to sit
    // out (the compiler waits for an atomic load at once).  The new count goes back at once: nobody else reads it during
    // this launch, and a store at the end would be the last thing the launch waits for.
    if (wv == 4 && own_epoch) {
        const uint32_t e = __hip_atomic_load(own_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
        if (lane == 0) {
            __hip_atomic_store(own_epoch, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            epoch_s = e;
        }
    }
    // ---- GEMM: this wave's share of the K steps
    const int steps = p.K / 32;
    const int s0 = wv < 4 ? steps * wv / 4 : 0, s1 = wv < 4 ? steps * (wv + 1) / 4 : 0;
    const bool rowok = row0 + n < p.T;
    const uint16_t* xrow = reinterpret_cast<const uint16_t*>(R.x) + (int64_t)(rowok ? row0 + n : 0) * R.x_row_stride + 8 * kg;
    const uint16_t* wrow[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
        wrow[j] = reinterpret_cast<const uint16_t*>(R.w) + (int64_t)(col0 + 16 * j + n) * R.w_row_stride + 8 * kg;
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // U k-steps per round: all 5 U loads of a round are in flight before its first MFMA (4 waves x 20 KiB per CU), which
    // is what keeps a CU's share of the weight stream moving - one k-step at a time the loop is pure memory latency
    constexpr int U = 4;
    for (int s = s0; s < s1; s += U) {
        u32x4 xb[U], wa[U][4];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k0 = 32 * (s + u < s1 ? s + u : s1 - 1);          // (past the end: a harmless re-read, zeroed below)
            xb[u] = *reinterpret_cast<const u32x4*>(xrow + k0);
#pragma unroll
            for (int j = 0; j < 4; ++j) wa[u][j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wrow[j] + k0));
        }
        __builtin_amdgcn_sched_barrier(0);                               // every load of the round is out before its first MFMA
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!rowok || s + u >= s1) xb[u] = u32x4{0u, 0u, 0u, 0u};
            const bf16x8 b = __builtin_bit_cast(bf16x8, xb[u]);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = mfma_16x16x32<F16>(__builtin_bit_cast(bf16x8, wa[u][j]), b, acc[j]);
        }
    }
    // acc[j][r] = partial out[row0 + n][col0 + 16 j + 4 kg + r]
    if (wv < 4)
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(&red[wv][n][16 * j + 4 * kg]) = acc[j];
    __syncthreads();
    if (own_epoch) epoch = epoch_s;

    // ---- push: thread (half, idx) holds row idx / 8, columns 8 (idx % 8) .. + 8; the two halves serve alternate peers
    const int half = threadIdx.x >> 7, idx = threadIdx.x & 127;
    const int row = idx >> 3, c8 = (idx & 7) * 8;
    const int parity = epoch & 1u;
    const bool live = row0 + row < p.T && threadIdx.x < 256;       // (wave 4 has no part in the push and the reduction)
    {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (red[0][row][c8 + e] + red[1][row][c8 + e]) + (red[2][row][c8 + e] + red[3][row][c8 + e]);
        const u32x4 val = {pack16<F16>(v[0], v[1]), pack16<F16>(v[2], v[3]), pack16<F16>(v[4], v[5]), pack16<F16>(v[6], v[7])};
        const int64_t off = ((int64_t)(parity * p.world + R.rank) * p.T_cap + row0 + row) * p.N + col0 + c8;
        if (live)
            for (int q = half; q < p.world; q += 2)
                st16_sys(reinterpret_cast<uint16_t*>(R.recv[q]) + off, val);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every piece of the tile has been acknowledged by its destination
    __syncthreads();                                     // ... for all 256 threads: only then the flags go out
    if ((int)threadIdx.x < p.world)
        __hip_atomic_store(R.flags[threadIdx.x] + (int64_t)R.rank * tiles_cap + tile, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);

    // ---- wait for this tile of every rank (bounded), then add the slots in rank order
    if ((int)threadIdx.x < p.world) {
        const uint32_t* f = R.flags[R.rank] + (int64_t)threadIdx.x * tiles_cap + tile;
        int spins = 0;
        while ((int32_t)(ld_sys(f) - epoch) < 0) {
            __builtin_amdgcn_s_sleep(4);
            if (++spins > kOprojPollLimit) {
                if (p.error) __hip_atomic_store(p.error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                gave_up_s = 1;                           // the tile's rows come out as NaN: a sum without every rank is not a result
                break;
            }
        }
    }
    __syncthreads();                                     // (the slots are read with cache-bypassing loads, after the flags)
    if (half == 0 && live) {
        float s[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] = 0.f;
        const uint16_t* mine = reinterpret_cast<const uint16_t*>(R.recv[R.rank]);
        for (int src = 0; src < p.world; ++src) {
            const int64_t off = ((int64_t)(parity * p.world + src) * p.T_cap + row0 + row) * p.N + col0 + c8;
            const u32x4 w = ld16_sys(mine + off);
#pragma unroll
            for (int e = 0; e < 4; ++e) { s[2 * e] += unpack16_lo<F16>(w[e]); s[2 * e + 1] += unpack16_hi<F16>(w[e]); }
        }
        u32x4 o = {pack16<F16>(s[0], s[1]), pack16<F16>(s[2], s[3]), pack16<F16>(s[4], s[5]), pack16<F16>(s[6], s[7])};
        // A peer never arrived (the wait above gave up): whatever sits in its slot is not its tile.  NaN the rows instead of
        // handing plausible numbers on - the model's next argmax / comparison then cannot look right - and leave *error set.
        // After that the instance is out of step with its peers for good (a late push can land in a half that a faster rank
        // has re-used): buffers, flags and epoch_state have to be zeroed again by every rank before the next call.
        if (gave_up_s) { const uint32_t nan2 = F16 ? 0x7e007e00u : 0x7fc07fc0u; o = u32x4{nan2, nan2, nan2, nan2}; }
        *reinterpret_cast<u32x4*>(reinterpret_cast<uint16_t*>(R.out) + (int64_t)(row0 + row) * R.out_row_stride + col0 + c8) = o;
    }
}

}  // namespace m355

using namespace m355;

// Freed allocations are kept, per device and size, for the next m355_comm_alloc of that size: on ROCm 7.2 / gfx950 a normal
// allocation that later lands on the address range of a hipFree'd UNCACHED one was seen read through stale 4 KiB
// translations (whole pages of a fresh tensor holding old data), so uncached memory is never handed back to the driver.
namespace {
struct CommPool {
    std::mutex mu;
    std::unordered_map<void*, std::pair<int, int64_t>> live;                   // ptr -> (device, bytes)
    std::map<std::pair<int, int64_t>, std::vector<void*>> spare;              // (device, bytes) -> freed blocks
};
CommPool& comm_pool() { static CommPool p; return p; }
}  // namespace

extern "C" int m355_comm_alloc(int64_t bytes, void** ptr) {
    if (!ptr || bytes <= 0) return M355_ERR_INVALID_ARG;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return M355_ERR_LAUNCH; }
    {
        CommPool& cp = comm_pool();
        std::lock_guard<std::mutex> lock(cp.mu);
        auto it = cp.spare.find({dev, bytes});
        if (it != cp.spare.end() && !it->second.empty()) {
            void* q = it->second.back();
            if (hipDeviceSynchronize() != hipSuccess || hipMemset(q, 0, (size_t)bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
                (void)hipGetLastError();
                return M355_ERR_LAUNCH;
            }
            it->second.pop_back();
            cp.live[q] = {dev, bytes};
            *ptr = q;
            return M355_OK;
        }
    }
    void* p = nullptr;
    if (hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocUncached) != hipSuccess || !p) { (void)hipGetLastError(); return M355_ERR_LAUNCH; }
    if (hipMemset(p, 0, (size_t)bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { (void)hipGetLastError(); (void)hipFree(p); return M355_ERR_LAUNCH; }
    {
        CommPool& cp = comm_pool();
        std::lock_guard<std::mutex> lock(cp.mu);
        cp.live[p] = {dev, bytes};
    }
    *ptr = p;
    return M355_OK;
}
extern "C" int m355_comm_zero(void* ptr, int64_t bytes) {
    if (!ptr || bytes <= 0) return M355_ERR_INVALID_ARG;
    if (hipMemset(ptr, 0, (size_t)bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { (void)hipGetLastError(); return M355_ERR_LAUNCH; }
    return M355_OK;
}
extern "C" int m355_comm_free(void* ptr) {             // (back to the pool, see above; never hipFree)
    if (!ptr) return M355_OK;
    CommPool& cp = comm_pool();
    std::lock_guard<std::mutex> lock(cp.mu);
    auto it = cp.live.find(ptr);
    if (it == cp.live.end()) return M355_ERR_INVALID_ARG;       // not one of ours (or freed twice)
    cp.spare[it->second].push_back(ptr);
    cp.live.erase(it);
    return M355_OK;
}
extern "C" int m355_comm_export(void* ptr, void* handle64) {
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "handle size");
    if (!ptr || !handle64) return M355_ERR_INVALID_ARG;
    if (hipIpcGetMemHandle(reinterpret_cast<hipIpcMemHandle_t*>(handle64), ptr) != hipSuccess) { (void)hipGetLastError(); return M355_ERR_LAUNCH; }
    return M355_OK;
}
extern "C" int m355_comm_open(const void* handle64, void** ptr) {
    if (!ptr || !handle64) return M355_ERR_INVALID_ARG;
    hipIpcMemHandle_t h;
    memcpy(&h, handle64, sizeof h);
    if (hipIpcOpenMemHandle(ptr, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); return M355_ERR_LAUNCH; }
    return M355_OK;
}
extern "C" int m355_comm_close(void* ptr) {
    if (!ptr) return M355_OK;
    if (hipIpcCloseMemHandle(ptr) != hipSuccess) { (void)hipGetLastError(); return M355_ERR_LAUNCH; }
    return M355_OK;
}

extern "C" int64_t m355_oproj_recv_bytes(int32_t world, int32_t T_cap, int32_t N) {
    if (world <= 0 || world > M355_MAX_RANKS || T_cap <= 0 || N <= 0) return 0;
    return 2ll * world * T_cap * N * 2;
}
extern "C" int64_t m355_oproj_flag_bytes(int32_t world, int32_t T_cap, int32_t N) {
    if (world <= 0 || world > M355_MAX_RANKS || T_cap <= 0 || N <= 0 || N % 64) return 0;
    return (int64_t)world * ((T_cap + 15) / 16) * (N / 64) * (int64_t)sizeof(uint32_t);
}

extern "C" int m355_oproj_allreduce(const M355OprojRank* ranks, int32_t n_ranks, int32_t world, int32_t T, int32_t N, int32_t K,
                                    int32_t T_cap, uint32_t epoch, uint32_t* epoch_state, int32_t* error, int32_t dtype,
                                    void* stream) {
    if (T == 0) return M355_OK;
    if (!ranks || n_ranks < 1 || world < 1 || world > M355_MAX_RANKS || n_ranks > world || T < 0 || T > T_cap || N <= 0 || K <= 0)
        return M355_ERR_INVALID_ARG;
    if (n_ranks != 1 && n_ranks != world) return M355_ERR_INVALID_ARG;
    if ((uintptr_t)epoch_state & 3) return M355_ERR_INVALID_ARG;
    if (N % 64 || K % 32) return M355_ERR_UNSUPPORTED;
    if (dtype != M355_DTYPE_BF16 && dtype != M355_DTYPE_F16) return M355_ERR_UNSUPPORTED;
    if ((int64_t)((T + 15) / 16) * (N / 64) * n_ranks > 65535ll * 16) return M355_ERR_INVALID_ARG;
    if (n_ranks == world && world > 1) {
        // every rank of the world in ONE launch on this device: a tile waits for the same tile of the other ranks, so all
        // workgroups of the launch have to be resident together or the low ranks' tiles wait for ranks that cannot start
        int per_cu = 0;
        const void* fn = dtype == M355_DTYPE_F16 ? reinterpret_cast<const void*>(&oproj_allreduce_kernel<true>)
                                                 : reinterpret_cast<const void*>(&oproj_allreduce_kernel<false>);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 320, 0) != hipSuccess || per_cu < 1) { (void)hipGetLastError(); per_cu = 1; }
        if ((int64_t)((T + 15) / 16) * (N / 64) * n_ranks > (int64_t)cu_count() * per_cu) return M355_ERR_INVALID_ARG;
    }
    OprojParams p;
    memset(&p, 0, sizeof p);
    for (int i = 0; i < n_ranks; ++i) {
        const M355OprojRank& r = ranks[i];
        if (!r.x || !r.w || !r.out || r.rank < 0 || r.rank >= world) return M355_ERR_INVALID_ARG;
        if (((uintptr_t)r.x | (uintptr_t)r.w | (uintptr_t)r.out) & 15) return M355_ERR_INVALID_ARG;
        if (r.x_row_stride < K || r.w_row_stride < K || r.out_row_stride < N || (r.x_row_stride % 8) || (r.w_row_stride % 8) ||
            (r.out_row_stride % 8))
            return M355_ERR_INVALID_ARG;
        for (int q = 0; q < world; ++q)
            if (!r.recv[q] || !r.flags[q] || ((uintptr_t)r.recv[q] & 15)) return M355_ERR_INVALID_ARG;
        p.r[i] = r;
    }
    p.world = world; p.T = T; p.N = N; p.K = K; p.T_cap = T_cap; p.epoch = epoch; p.epoch_state = epoch_state; p.error = error;
    const dim3 grid((unsigned)(N / 64), (unsigned)((T + 15) / 16), (unsigned)n_ranks);
    if (dtype == M355_DTYPE_F16) hipLaunchKernelGGL((oproj_allreduce_kernel<true>), grid, dim3(320), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((oproj_allreduce_kernel<false>), grid, dim3(320), 0, (hipStream_t)stream, p);
    return check_launch();
}
