// Dev: what one LDS-DMA instruction costs the issuing wave, and whether the instruction's immediate offset moves the
// LDS destination (so that eight pieces can share one M0 value).
//   variant 0: M0 rewritten before every piece (s_mov m0 + s_nop + global_load_lds_dwordx4)
//   variant 1: one M0 write, eight pieces told apart by offset:-4096 ... offset:3072 (global address compensated)
// Each wave issues 8 pieces per round into its own 8 KiB of LDS from an L2-resident 8 KiB source, R rounds, with
// vmcnt(0) between rounds; cycles per round by s_memtime.  WPS waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int VAR>
__global__ __launch_bounds__(256) void k(const uint8_t* __restrict__ src, int rounds, unsigned long long* cyc, uint32_t* chk) {
    extern __shared__ __attribute__((aligned(1024))) uint8_t smem[];
    const int wid = threadIdx.x >> 6, l = threadIdx.x & 63;
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)smem + wid * 8192);
    const uint8_t* g = src + (size_t)(blockIdx.x & 63) * 8192 + l * 16;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < rounds; ++r) {
        if (VAR == 0) {
            uint32_t keep;
#define P(i) "s_mov_b32 m0, %" #i "\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[a" #i "], off\n\t"
            asm volatile("s_mov_b32 %0, m0\n\t"
                         "s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %9, off\n\t"
                         "s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %10, off\n\t"
                         "s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %11, off\n\t"
                         "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %12, off\n\t"
                         "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %13, off\n\t"
                         "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %14, off\n\t"
                         "s_mov_b32 m0, %7\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %15, off\n\t"
                         "s_mov_b32 m0, %8\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %16, off\n\t"
                         "s_mov_b32 m0, %0"
                         : "=&s"(keep) : "s"(lds0), "s"(lds0 + 1024), "s"(lds0 + 2048), "s"(lds0 + 3072), "s"(lds0 + 4096),
                           "s"(lds0 + 5120), "s"(lds0 + 6144), "s"(lds0 + 7168),
                           "v"(g), "v"(g + 1024), "v"(g + 2048), "v"(g + 3072), "v"(g + 4096), "v"(g + 5120), "v"(g + 6144), "v"(g + 7168) : "memory");
        } else {
            uint32_t keep;
            const uint8_t* gm = g + 4096;   // the immediate is added to the global address too
            asm volatile("s_mov_b32 %0, m0\n\t"
                         "s_mov_b32 m0, %1\n\ts_nop 0\n\t"
                         "global_load_lds_dwordx4 %2, off offset:-4096\n\t"
                         "global_load_lds_dwordx4 %2, off offset:-3072\n\t"
                         "global_load_lds_dwordx4 %2, off offset:-2048\n\t"
                         "global_load_lds_dwordx4 %2, off offset:-1024\n\t"
                         "global_load_lds_dwordx4 %2, off\n\t"
                         "global_load_lds_dwordx4 %2, off offset:1024\n\t"
                         "global_load_lds_dwordx4 %2, off offset:2048\n\t"
                         "global_load_lds_dwordx4 %2, off offset:3072\n\t"
                         "s_mov_b32 m0, %0"
                         : "=&s"(keep) : "s"(lds0 + 4096), "v"(gm) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    if (l == 0) cyc[blockIdx.x * 4 + wid] = t1 - t0;
    // checksum of what landed: bytes of piece j must equal the source's piece j
    if (blockIdx.x == 0 && wid == 0) {
        uint32_t bad = 0;
        for (int j = 0; j < 8; ++j) {
            const uint32_t got = *reinterpret_cast<const uint32_t*>(smem + j * 1024 + l * 16);
            const uint32_t want = *reinterpret_cast<const uint32_t*>(src + j * 1024 + l * 16);
            bad |= (got != want) << j;
        }
        atomicOr(chk, bad);
    }
}

int main() {
    uint8_t* src; unsigned long long* cyc; uint32_t* chk;
    const size_t n = 64 * 8192;
    CK(hipMalloc(&src, n)); CK(hipMalloc(&cyc, 8192 * 8)); CK(hipMalloc(&chk, 4));
    std::vector<uint32_t> h(n / 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (uint32_t)(i * 2654435761u);
    CK(hipMemcpy(src, h.data(), n, hipMemcpyHostToDevice));
    const int rounds = 200;
    for (int wps = 1; wps <= 2; ++wps)
        for (int var = 0; var < 2; ++var) {
            CK(hipMemset(chk, 0, 4));
            const int grid = 256 * wps;
            for (int rep = 0; rep < 2; ++rep) {
                if (var == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 32768, 0, src, rounds, cyc, chk);
                else hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 32768, 0, src, rounds, cyc, chk);
                CK(hipDeviceSynchronize());
            }
            std::vector<unsigned long long> c(grid * 4);
            uint32_t bad;
            CK(hipMemcpy(c.data(), cyc, c.size() * 8, hipMemcpyDeviceToHost));
            CK(hipMemcpy(&bad, chk, 4, hipMemcpyDeviceToHost));
            double s = 0; for (auto x : c) s += (double)x;
            printf("waves/SIMD %d variant %d: %.0f cycles per round of 8 pieces (%.1f per piece), landed-wrong mask 0x%02x\n",
                   wps, var, s / c.size() / rounds, s / c.size() / rounds / 8, bad);
        }
    return 0;
}
