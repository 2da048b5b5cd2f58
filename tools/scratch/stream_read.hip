// Dev: streaming-read ceiling of the box.  Plain dwordx4 loads and LDS-DMA, contiguous and strided-row patterns.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// each wave streams a contiguous span; U loads in flight per lane
template <int U>
__global__ __launch_bounds__(256) void read_plain(const u32x4* __restrict__ src, size_t n16, uint32_t* out) {
    const size_t nthreads = (size_t)gridDim.x * blockDim.x;
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    u32x4 acc = {0, 0, 0, 0};
    for (size_t i = tid; i + (U - 1) * nthreads < n16; i += U * nthreads) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = src[i + u * nthreads];
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u];
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[0] = 1;
}

// per-wave private contiguous ranges (like the decode kernel's token ranges), LDS-DMA, R KiB in flight per wave
template <int R>
__global__ __launch_bounds__(64) void read_dma(const uint8_t* __restrict__ src, size_t bytes_per_wave, uint32_t* out) {
    extern __shared__ uint8_t smem[];
    const uint8_t* base = src + (size_t)blockIdx.x * bytes_per_wave + threadIdx.x * 16;
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)smem);
    const size_t n = bytes_per_wave / 1024;
    size_t issued = 0;
    auto issue = [&](int slot) {
        uint32_t keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(base + issued * 1024), "s"(lds0 + slot * 1024) : "memory");
        ++issued;
    };
    for (int j = 0; j < R && issued < n; ++j) issue(j);
    for (size_t c = 0; c < n; c += R) {
#pragma unroll
        for (int j = 0; j < R; ++j) {
            if (c + j < n) {
                asm volatile("s_waitcnt vmcnt(%0)" :: "i"(R - 1 > 63 ? 63 : R - 1) : "memory");
                if (issued < n) issue(j);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (smem[threadIdx.x] == 0x77 && bytes_per_wave == 3) out[0] = 1;
}

// Paged patterns.  `pages` = shuffled list of 16 KiB page indices; wave w owns pages [w*ppw, (w+1)*ppw).
// MODE 0: a DMA instruction reads 1 KiB contiguous (16 per page).  MODE 1 ("head column"): the wave reads only the
// 128-byte column `h` of each of the page's 16 rows of 1 KiB - 2 instructions of 8 rows per page - like one kv head.
template <int R, int MODE>
__global__ __launch_bounds__(64) void read_paged(const uint8_t* __restrict__ src, const int* __restrict__ pages, int ppw,
                                                 int nheads, uint32_t* out) {
    extern __shared__ uint8_t smem[];
    const int lane = threadIdx.x;
    const int w = MODE == 1 ? blockIdx.x / nheads : blockIdx.x;
    const int h = MODE == 1 ? (blockIdx.x + w / 32) % nheads : 0;
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)smem);
    const int ipp = MODE == 1 ? 2 : 16;                  // instructions per page
    const int n = ppw * ipp;
    int issued = 0;
    auto issue = [&](int slot) {
        const int pg = pages[w * ppw + issued / ipp];
        const int i = issued % ipp;
        const uint8_t* a = MODE == 1 ? src + (size_t)pg * 16384 + (size_t)(i * 8 + lane / 8) * 1024 + h * 128 + (lane % 8) * 16
                                     : src + (size_t)pg * 16384 + i * 1024 + lane * 16;
        uint32_t keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(a), "s"(lds0 + slot * 1024) : "memory");
        ++issued;
    };
    for (int j = 0; j < R && issued < n; ++j) issue(j);
    for (int c = 0; c < n; c += R) {
#pragma unroll
        for (int j = 0; j < R; ++j) {
            if (c + j < n) {
                asm volatile("s_waitcnt vmcnt(%0)" :: "i"(R - 1 > 63 ? 63 : R - 1) : "memory");
                if (issued < n) issue(j);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (smem[threadIdx.x] == 0x77 && ppw == 3) out[0] = 1;
}

// Stage-wise ring like the decode kernel: groups of GI instructions; wait for the older group, then issue a new one.
// pages of PB bytes, shuffled; one "pool" or two pools (K and V) interleaved.
template <int GI, int PB, int POOLS>
__global__ __launch_bounds__(64) void read_staged(const uint8_t* __restrict__ src, const int* __restrict__ pages, int ppw, size_t pool_stride, uint32_t* out) {
    extern __shared__ uint8_t smem[];
    const int lane = threadIdx.x;
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)smem);
    constexpr int IPP = PB / 1024;                        // instructions per page (per pool)
    const int n = ppw * IPP / (GI / POOLS);               // stages
    int issued = 0;
    auto issue_stage = [&](int slot) {
#pragma unroll
        for (int j = 0; j < GI / POOLS; ++j) {
            const int q = issued * (GI / POOLS) + j;
            const int pg = pages[blockIdx.x * ppw + q / IPP];
            const uint8_t* a = src + (size_t)pg * PB + (q % IPP) * 1024 + lane * 16;
#pragma unroll
            for (int pl = 0; pl < POOLS; ++pl) {
                uint32_t keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(a + pl * pool_stride), "s"(lds0 + (slot * GI + j * POOLS + pl) * 1024) : "memory");
            }
        }
        ++issued;
    };
    issue_stage(0);
    for (int c = 0; c < n; ++c) {
        if (issued < n) issue_stage(issued & 1);
        if (c + 1 < n) asm volatile("s_waitcnt vmcnt(%0)" :: "i"(GI) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (smem[threadIdx.x] == 0x77 && ppw == 3) out[0] = 1;
}

int main() {
    const size_t bytes = (size_t)2400 << 20;
    uint8_t* d; uint32_t* o;
    CK(hipMalloc(&d, bytes)); CK(hipMalloc(&o, 4));
    CK(hipMemset(d, 1, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, auto launch) {
        launch(); (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        for (int i = 0; i < 5; ++i) launch();
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        printf("%-44s %7.3f ms  %6.2f TB/s\n", name, ms, bytes / (ms * 1e-3) / 1e12);
    };
    const size_t n16 = bytes / 16;
    timeit("plain dwordx4 U=4  grid 256x8 blocks", [&] { read_plain<4><<<2048, 256>>>((const u32x4*)d, n16, o); });
    timeit("plain dwordx4 U=8  grid 256x8 blocks", [&] { read_plain<8><<<2048, 256>>>((const u32x4*)d, n16, o); });
    timeit("plain dwordx4 U=8  grid 256x16 blocks", [&] { read_plain<8><<<4096, 256>>>((const u32x4*)d, n16, o); });
    timeit("plain dwordx4 U=16 grid 256x4 blocks", [&] { read_plain<16><<<1024, 256>>>((const u32x4*)d, n16, o); });
    for (int waves : {1024, 2048, 4096}) {
        const size_t bpw = bytes / waves / 1024 * 1024;
        char nm[96];
        snprintf(nm, sizeof nm, "LDS-DMA R=8  KiB in flight, %d waves", waves);
        timeit(nm, [&] { read_dma<8><<<waves, 64, 8 * 1024>>>(d, bpw, o); });
        snprintf(nm, sizeof nm, "LDS-DMA R=16 KiB in flight, %d waves", waves);
        timeit(nm, [&] { read_dma<16><<<waves, 64, 16 * 1024>>>(d, bpw, o); });
        snprintf(nm, sizeof nm, "LDS-DMA R=32 KiB in flight, %d waves", waves);
        timeit(nm, [&] { read_dma<32><<<waves, 64, 32 * 1024>>>(d, bpw, o); });
    }
    {   // paged patterns over the same buffer
        const int npages = (int)(bytes / 16384);
        std::vector<int> hp(npages);
        for (int i = 0; i < npages; ++i) hp[i] = i;
        uint64_t st = 88172645463325252ull;
        for (int i = npages - 1; i > 0; --i) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; int j = (int)(st % (uint64_t)(i + 1)); int t = hp[i]; hp[i] = hp[j]; hp[j] = t; }
        int* dp; CK(hipMalloc(&dp, npages * sizeof(int)));
        CK(hipMemcpy(dp, hp.data(), npages * sizeof(int), hipMemcpyHostToDevice));
        const int waves = 2048;
        timeit("paged shuffled 16K pages, 1 KiB rows, R=16", [&] { read_paged<16, 0><<<waves, 64, 16 * 1024>>>(d, dp, npages / waves, 1, o); });
        // head-column: 8 heads x 256 ranges; each (w,h) wave reads column h of pages of range w
        timeit("paged head-column 8x128B per instr, R=16", [&] { read_paged<16, 1><<<256 * 8, 64, 16 * 1024>>>(d, dp, npages / 256, 8, o); });
        timeit("paged head-column 8x128B per instr, R=8", [&] { read_paged<8, 1><<<256 * 8, 64, 8 * 1024>>>(d, dp, npages / 256, 8, o); });
        // the decode kernel's own size: 2 x 75.8 MB (K and V of one layer) = 9254 pages, 36 per (range, head) wave, cold
        // (each timed launch reads a different 151 MB window of the 2.4 GB buffer)
        {
            const int ppw = 36;
            int win = 0;
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0);
            for (int i = 0; i < 10; ++i) { read_paged<16, 1><<<256 * 8, 64, 16 * 1024>>>(d, dp + (win % 12) * 256 * ppw, ppw, 8, o); ++win; }
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 10;
            printf("%-44s %7.3f us  %6.2f TB/s\n", "head-column, 151 MB per launch (decode size)", ms * 1e3, 256.0 * ppw * 16384 / (ms * 1e-3) / 1e12);
        }
        {   // staged rings: 1024 waves (4 per CU); 4 KiB pages, one pool / two pools half a buffer apart
            const int np4 = (int)(bytes / 2 / 4096);      // pages per pool when two pools share the buffer
            std::vector<int> h4(np4);
            for (int i = 0; i < np4; ++i) h4[i] = i;
            for (int i = np4 - 1; i > 0; --i) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; int j = (int)(st % (uint64_t)(i + 1)); int t = h4[i]; h4[i] = h4[j]; h4[j] = t; }
            int* d4; CK(hipMalloc(&d4, np4 * sizeof(int)));
            CK(hipMemcpy(d4, h4.data(), np4 * sizeof(int), hipMemcpyHostToDevice));
            const int waves = 1024;
            auto t2 = [&](const char* name, auto launch, double nbytes) {
                launch(); (void)hipDeviceSynchronize();
                (void)hipEventRecord(e0);
                for (int i = 0; i < 5; ++i) launch();
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
                printf("%-52s %7.3f ms  %6.2f TB/s\n", name, ms, nbytes / (ms * 1e-3) / 1e12);
            };
            const int ppw = np4 / waves;
            t2("staged 16 instr/stage, 4K pages, 2 pools, 1024 waves", [&] { read_staged<16, 4096, 2><<<waves, 64, 32 * 1024>>>(d, d4, ppw, bytes / 2, o); }, 2.0 * waves * ppw * 4096);
            t2("staged  8 instr/stage, 4K pages, 2 pools, 1024 waves", [&] { read_staged<8, 4096, 2><<<waves, 64, 16 * 1024>>>(d, d4, ppw, bytes / 2, o); }, 2.0 * waves * ppw * 4096);
            t2("staged 16 instr/stage, 4K pages, 1 pool,  1024 waves", [&] { read_staged<16, 4096, 1><<<waves, 64, 32 * 1024>>>(d, d4, ppw, 0, o); }, 1.0 * waves * ppw * 4096);
            t2("staged 16 instr/stage, 4K pages, 2 pools, 2048 waves", [&] { read_staged<16, 4096, 2><<<2048, 64, 32 * 1024>>>(d, d4, np4 / 2048, bytes / 2, o); }, 2.0 * 2048 * (np4 / 2048) * 4096);
        }
        for (int i = 0; i < npages; ++i) hp[i] = i;
        CK(hipMemcpy(dp, hp.data(), npages * sizeof(int), hipMemcpyHostToDevice));
        timeit("head-column, pages in order, R=16", [&] { read_paged<16, 1><<<256 * 8, 64, 16 * 1024>>>(d, dp, npages / 256, 8, o); });
    }
    return 0;
}
