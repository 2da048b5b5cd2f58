// Dev: is the per-XCD streaming-rate asymmetry a property of the physical XCD?  Per-wave durations of
// bare LDS-DMA streaming kernels grouped by the hardware XCC_ID, over several launches.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

struct Stamp { unsigned long long t0, t1; uint32_t xcc, pad; };

// MODE 0: contiguous 1 KiB per instruction, wave-private contiguous span.  MODE 1: head column (8 rows x 128 B) of
// shuffled 16 KiB pages, heads rotated over the XCDs like the decode kernel.
template <int R, int MODE, int NT = 0>
__global__ __launch_bounds__(64) void stream(const uint8_t* __restrict__ src, const int* __restrict__ pages, int ppw,
                                             int nheads, Stamp* st) {
    extern __shared__ uint8_t smem[];
    const int lane = threadIdx.x;
    const int w = MODE == 1 ? blockIdx.x / nheads : blockIdx.x;
    const int h = MODE == 1 ? (blockIdx.x + w / 32) % nheads : 0;
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)smem);
    const int ipp = MODE == 1 ? 2 : 16;
    const int n = ppw * ipp;
    int issued = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    auto issue = [&](int slot) {
        const int pg = MODE == 1 ? pages[w * ppw + issued / ipp] : w * ppw + issued / ipp;
        const int i = issued % ipp;
        const uint8_t* a = MODE == 1 ? src + (size_t)pg * 16384 + (size_t)(i * 8 + lane / 8) * 1024 + h * 128 + (lane % 8) * 16
                                     : src + (size_t)pg * 16384 + i * 1024 + lane * 16;
        uint32_t keep;
        if constexpr (NT) asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(a), "s"(lds0 + slot * 1024) : "memory");
        else asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(a), "s"(lds0 + slot * 1024) : "memory");
        ++issued;
    };
    for (int j = 0; j < R && issued < n; ++j) issue(j);
    for (int c = 0; c < n; c += R) {
#pragma unroll
        for (int j = 0; j < R; ++j) {
            if (c + j < n) {
                asm volatile("s_waitcnt vmcnt(%0)" :: "i"(R - 1) : "memory");
                if (issued < n) issue(j);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) {
        st[blockIdx.x].t0 = t0; st[blockIdx.x].t1 = t1;
        st[blockIdx.x].xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15;      // HW_REG_XCC_ID[3:0]
    }
}

int main() {
    const size_t bytes = (size_t)2400 << 20;
    uint8_t* d; CK(hipMalloc(&d, bytes)); CK(hipMemset(d, 1, bytes));
    const int npages = (int)(bytes / 16384);
    std::vector<int> hp(npages);
    for (int i = 0; i < npages; ++i) hp[i] = i;
    uint64_t s = 88172645463325252ull;
    for (int i = npages - 1; i > 0; --i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; int j = (int)(s % (uint64_t)(i + 1)); std::swap(hp[i], hp[j]); }
    int* dp; CK(hipMalloc(&dp, npages * sizeof(int)));
    CK(hipMemcpy(dp, hp.data(), npages * sizeof(int), hipMemcpyHostToDevice));
    const int waves = 2048;
    Stamp* ds; CK(hipMalloc(&ds, waves * sizeof(Stamp)));
    std::vector<Stamp> hs(waves);
    auto report = [&](const char* name) {
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(hs.data(), ds, waves * sizeof(Stamp), hipMemcpyDeviceToHost);
        unsigned long long first = ~0ull, last = 0;
        for (auto& x : hs) { first = std::min(first, x.t0); last = std::max(last, x.t1); }
        printf("%s: kernel span %.2f us\n", name, (last - first) * 0.01);
        for (int xcc = 0; xcc < 8; ++xcc) {
            std::vector<double> dur, end; int labels[8] = {0};
            for (int b = 0; b < waves; ++b) if ((int)hs[b].xcc == xcc) { dur.push_back((hs[b].t1 - hs[b].t0) * 0.01); end.push_back((hs[b].t1 - first) * 0.01); ++labels[b % 8]; }
            if (dur.empty()) continue;
            std::sort(dur.begin(), dur.end()); std::sort(end.begin(), end.end());
            int lab = (int)(std::max_element(labels, labels + 8) - labels);
            printf("  xcc %d (label %d, %4zu waves): wave time median %7.2f us  max %7.2f   last end %7.2f us\n", xcc, lab, dur.size(), dur[dur.size() / 2], dur.back(), end.back());
        }
    };
    for (int rep = 0; rep < 3; ++rep) {
        // decode-size launches: 36 pages per (range, head) wave = 151 MB; a different window each time (cold)
        stream<16, 1><<<waves, 64, 16 * 1024>>>(d, dp + rep * 256 * 36, 36, 8, ds);
        report("head-column 151 MB");
    }
    for (int rep = 0; rep < 2; ++rep) {
        stream<16, 1><<<waves, 64, 16 * 1024>>>(d, dp, npages / 256, 8, ds);
        report("head-column 2.4 GB");
    }
    for (int rep = 0; rep < 2; ++rep) {
        stream<16, 0><<<waves, 64, 16 * 1024>>>(d, dp, npages / waves, 1, ds);
        report("contiguous 2.4 GB");
    }
    // non-temporal loads (what the decode kernel uses since the end of round 1)
    for (int rep = 0; rep < 2; ++rep) {
        stream<16, 1, 1><<<waves, 64, 16 * 1024>>>(d, dp, npages / 256, 8, ds);
        report("head-column 2.4 GB, nt");
    }
    for (int rep = 0; rep < 2; ++rep) {
        stream<16, 0, 1><<<waves, 64, 16 * 1024>>>(d, dp, npages / waves, 1, ds);
        report("contiguous 2.4 GB, nt");
    }
    for (int rep = 0; rep < 3; ++rep) {
        stream<16, 1, 1><<<waves, 64, 16 * 1024>>>(d, dp + (4 + rep) * 256 * 36, 36, 8, ds);
        report("head-column 151 MB, nt");
    }
    return 0;
}
