// Hardware probes behind the round-2 decode kernel (decode_share.hip).  Build: hipcc --offload-arch=gfx950 -O2 probe_r2.hip -o probe_r2
//   1. LDS-DMA (global_load_lds_dword / dwordx4) with a partial EXEC mask: do inactive lanes leave LDS alone?
//   2. one workgroup of 576 threads with ~156 KiB of dynamic LDS: does it launch?
//   3. sc1 publish / sc1 poll hand-off between workgroups on different XCDs, consumer's lines L2-warm from a previous
//      launch (the hand-off the share kernel uses for requests cut by a share boundary).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void glds4(const void* gsrc, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

__global__ void probe_exec(const uint32_t* src, uint32_t* out, int n16, int n4) {
    __shared__ __attribute__((aligned(1024))) uint32_t lds[512];
    const int lane = threadIdx.x;
    for (int i = lane; i < 512; i += 64) lds[i] = 0xdeadbeefu;
    __syncthreads();
    const uint32_t base = (uint32_t)(uintptr_t)lds;
    if (lane < n16) glds16(src + lane * 4, base);                     // lanes [0, n16): 16 bytes each at base + 16*lane
    if (lane < n4) glds4(src + 256 + lane, base + 1024);              // lanes [0, n4): 4 bytes each at base + 1024 + 4*lane
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 512; i += 64) out[i] = lds[i];
}

__global__ __launch_bounds__(576) void probe_biglds(uint32_t* out, int words) {
    extern __shared__ __attribute__((aligned(1024))) uint32_t big[];
    for (int i = threadIdx.x; i < words; i += blockDim.x) big[i] = i * 2654435761u;
    __syncthreads();
    uint32_t acc = 0;
    for (int i = threadIdx.x; i < words; i += blockDim.x) acc ^= big[words - 1 - i];
    atomicXor(out, acc);
}

// hand-off: block b (b >= 1) waits for block b-1's record; block b-1 publishes after `delay` spins.  Records are 1 KiB
// (256 dwords) + flag.  Run twice with different values so the consumer's L2 may hold the first launch's lines.
__global__ __launch_bounds__(64) void probe_handoff(uint32_t* rec, uint32_t* flags, uint32_t* out, uint32_t val, int delay, uint32_t* timeouts) {
    const int b = blockIdx.x, lane = threadIdx.x;
    // publish own record
    for (volatile int i = 0; i < delay * (1 + (b % 3)); ++i) {}
    uint32_t* mine = rec + (size_t)b * 256;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 v = {val + b, val + b + lane, val ^ (uint32_t)lane, val * 3u + b};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(mine + lane * 4), "v"(v) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_store(flags + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (b == 0) return;
    // consume the predecessor's
    int spins = 0;
    uint32_t f = 0;
    if (lane == 0) {
        while ((f = __hip_atomic_load(flags + b - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0 && spins < (1 << 22)) { __builtin_amdgcn_s_sleep(8); ++spins; }
        if (f == 0) atomicAdd(timeouts, 1u);
    }
    f = __shfl(f, 0);
    const uint32_t* theirs = rec + (size_t)(b - 1) * 256;
    u32x4 got;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(got) : "v"(theirs + lane * 4) : "memory");
    const uint32_t bb = b - 1;
    const bool ok = got[0] == val + bb && got[1] == val + bb + lane && got[2] == (val ^ (uint32_t)lane) && got[3] == val * 3u + bb;
    if (!ok) atomicAdd(out, 1u);
    if (lane == 0) __hip_atomic_store(flags + b - 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // consumer resets
}

int main() {
    // ---- 1
    std::vector<uint32_t> h(512);
    for (int i = 0; i < 512; ++i) h[i] = 0x1000 + i;
    uint32_t *src, *out;
    CK(hipMalloc(&src, 2048)); CK(hipMalloc(&out, 2048));
    CK(hipMemcpy(src, h.data(), 2048, hipMemcpyHostToDevice));
    for (int n16 : {64, 32, 8}) for (int n4 : {64, 40}) {
        probe_exec<<<1, 64>>>(src, out, n16, n4);
        CK(hipDeviceSynchronize());
        std::vector<uint32_t> r(512);
        CK(hipMemcpy(r.data(), out, 2048, hipMemcpyDeviceToHost));
        int bad16 = 0, bad4 = 0;
        for (int i = 0; i < 256; ++i) { const uint32_t want = i < n16 * 4 ? h[i] : 0xdeadbeefu; bad16 += r[i] != want; }
        for (int i = 0; i < 64; ++i) { const uint32_t want = i < n4 ? h[256 + i] : 0xdeadbeefu; bad4 += r[256 + i] != want; }
        printf("exec-masked LDS-DMA: %2d lanes x16B -> %s (%d wrong), %2d lanes x4B -> %s (%d wrong)\n", n16, bad16 ? "FAIL" : "ok", bad16,
               n4, bad4 ? "FAIL" : "ok", bad4);
    }
    // ---- 2
    for (int kib : {64, 128, 152, 156, 160}) {
        const int bytes = kib * 1024;
        hipError_t e = hipFuncSetAttribute((const void*)probe_biglds, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        CK(hipMemset(out, 0, 4));
        if (e == hipSuccess) { probe_biglds<<<2, 576, bytes>>>(out, bytes / 4); e = hipGetLastError(); }
        hipError_t e2 = hipDeviceSynchronize();
        printf("dynamic LDS %3d KiB, 576 threads: setattr/launch %s, sync %s\n", kib, hipGetErrorString(e), hipGetErrorString(e2));
        if (e2 != hipSuccess) return 3;
    }
    // ---- 3
    const int NB = 256;
    uint32_t *rec, *flags, *bad, *tmo;
    CK(hipMalloc(&rec, NB * 1024)); CK(hipMalloc(&flags, NB * 4)); CK(hipMalloc(&bad, 4)); CK(hipMalloc(&tmo, 4));
    CK(hipMemset(flags, 0, NB * 4)); CK(hipMemset(bad, 0, 4)); CK(hipMemset(tmo, 0, 4)); CK(hipMemset(rec, 0, NB * 1024));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 200; ++rep) {
        if (rep == 100) CK(hipEventRecord(e0));
        probe_handoff<<<NB, 64>>>(rec, flags, bad, 0x5000u + 977u * rep, (rep % 4) * 2000, tmo);
    }
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    uint32_t hb = 0, ht = 0; std::vector<uint32_t> hf(NB);
    CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&ht, tmo, 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hf.data(), flags, NB * 4, hipMemcpyDeviceToHost));
    int left = 0; for (int i = 0; i < NB - 1; ++i) left += hf[i] != 0;
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("sc1 hand-off, 200 launches x 255 edges: %u stale lanes, %u poll timeouts, %d flags left raised (want 0 0 0), %.1f us/launch\n", hb, ht, left, ms * 10.f);
    return (hb || ht || left) ? 4 : 0;
}
