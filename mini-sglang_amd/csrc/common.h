// Shared device helpers for libmi355attn (gfx950 only: 64-lane waves, v_dot2c_f32_bf16, GFX9 DPP).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <atomic>

#include "../../include/mi355attn.h"

namespace m355 {

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kWave = 64;
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kNegBig = -1.0e30f;   // finite stand-in for -inf: (m_old - m_new) never makes a NaN

__device__ __forceinline__ float exp2_fast(float x) { return __builtin_amdgcn_exp2f(x); }

// acc += a.lo*b.lo + a.hi*b.hi with bf16 pairs packed in a dword (v_dot2c_f32_bf16)
__device__ __forceinline__ float dot2_bf16(uint32_t a, uint32_t b, float acc) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a), __builtin_bit_cast(bf16x2_t, b), acc, false);
}

__device__ __forceinline__ float bf16_lo(uint32_t w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t w) { return __builtin_bit_cast(float, w & 0xffff0000u); }

// f32 -> bf16 round-to-nearest-even through the hardware convert (keeps NaN a NaN)
// (the two-element vector convert is ONE v_cvt_pk_bf16_f32; two scalar casts cost a convert each plus a shift and an or)
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2v_t __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{lo, hi}, bf16x2v_t));
}

// ---- the 16-bit element type of q / K / V / out as a template flag: bf16 (F16 = false) or IEEE half (F16 = true).
// Fragments travel as bf16x8 (a 128-bit container either way); only the matrix instruction, the f32 -> 16-bit
// packing and the 16-bit -> f32 unpacking differ.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <bool F16>
__device__ __forceinline__ f32x4 mfma_16x16x32(bf16x8 a, bf16x8 b, f32x4 c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
template <bool F16>
__device__ __forceinline__ f32x16 mfma_32x32x16(bf16x8 a, bf16x8 b, f32x16 c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
template <bool F16>
__device__ __forceinline__ uint32_t pack16(float lo, float hi) {        // round to nearest even, one v_cvt_pk_*
    if constexpr (F16) {
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{lo, hi}, f16x2_t));
    } else {
        return pack_bf16(lo, hi);
    }
}
template <bool F16>
__device__ __forceinline__ float unpack16_lo(uint32_t w) {
    if constexpr (F16) return (float)__builtin_bit_cast(_Float16, (uint16_t)(w & 0xffffu));
    else return bf16_lo(w);
}
template <bool F16>
__device__ __forceinline__ float unpack16_hi(uint32_t w) {
    if constexpr (F16) return (float)__builtin_bit_cast(_Float16, (uint16_t)(w >> 16));
    else return bf16_hi(w);
}

// x + (x from the lane selected by a GFX9 DPP control); fuses into v_add_f32_dpp
template <int CTRL>
__device__ __forceinline__ float dpp_add(float x) {
    int r = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true);
    return x + __builtin_bit_cast(float, r);
}
constexpr int kDppXor1 = 0xB1;        // quad_perm:[1,0,3,2]
constexpr int kDppXor2 = 0x4E;        // quad_perm:[2,3,0,1]
constexpr int kDppHalfMirror = 0x141; // lane i <- lane 7-i within each 8
constexpr int kDppMirror = 0x140;     // lane i <- lane 15-i within each 16

// all-reduce (sum) over aligned groups of LANES consecutive lanes, LANES in {8, 16}
template <int LANES>
__device__ __forceinline__ float group_sum(float x) {
    x = dpp_add<kDppXor1>(x);
    x = dpp_add<kDppXor2>(x);
    x = dpp_add<kDppHalfMirror>(x);
    if constexpr (LANES == 16) x = dpp_add<kDppMirror>(x);
    return x;
}

// (own, partner) of a value across lanes l and l ^ MASK, MASK in {8, 16, 32}, in VALU only (no LDS round trip):
// DPP row rotate for 8, v_permlane16_swap / v_permlane32_swap for 16 / 32.  For the swaps which of a / b is "own"
// depends on the lane, so callers must use them symmetrically - and the same way for every value of one state.
template <int MASK>
__device__ __forceinline__ void lane_pair(float x, float& a, float& b) {
    const unsigned ux = __builtin_bit_cast(unsigned, x);
    if constexpr (MASK == 8) {
        a = x;
        b = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, (int)ux, 0x128 /*row_ror:8*/, 0xF, 0xF, true));
    } else if constexpr (MASK == 16) {
        // inline asm: with both inputs equal, hipcc (ROCm 7.2) folds the builtin's two results into one register
        unsigned ua = ux, ub = ux;
        asm volatile("v_nop\n\tv_nop\n\tv_permlane16_swap_b32 %0, %1" : "+v"(ua), "+v"(ub));
        a = __builtin_bit_cast(float, ua); b = __builtin_bit_cast(float, ub);
    } else {
        static_assert(MASK == 32, "lane_pair: unsupported distance");
        unsigned ua = ux, ub = ux;
        asm volatile("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %1" : "+v"(ua), "+v"(ub));
        a = __builtin_bit_cast(float, ua); b = __builtin_bit_cast(float, ub);
    }
}

// LDS-DMA: 16 bytes per lane from global memory straight into LDS at (lds_dst + 16*lane), no VGPR round trip.
// hipcc does not count these loads in its own s_waitcnt bookkeeping: callers wait with wait_vmcnt<N>() themselves.
__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// The same with a cache policy on the load.  POL 1 = `nt` (non-temporal): a stream that is read once - decode's K/V -
// then does not displace what L2 / Infinity Cache hold and does not pay for being kept (decode: 35.3 -> 31.4 us).
// 0 = default policy; 2..4 are measurement variants.
template <int POL>
__device__ __forceinline__ void glds16p(const void* gsrc, uint32_t lds_dst) {
    uint32_t keep;
#define M355_GLDS(MODS) asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" MODS \
                                     "\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory")
    if constexpr (POL == 1) M355_GLDS(" nt");
    else if constexpr (POL == 2) M355_GLDS(" sc1 nt");
    else if constexpr (POL == 3) M355_GLDS(" sc0 sc1 nt");
    else if constexpr (POL == 4) M355_GLDS(" sc0 nt");
    else if constexpr (POL == 5) M355_GLDS(" sc1");
    else if constexpr (POL == 6) M355_GLDS(" sc0 sc1");
    else M355_GLDS("");
#undef M355_GLDS
}
// N LDS-DMA loads back to back with ONE save / restore of M0 (each load still needs its own M0 = LDS destination)
template <int N>
__device__ __forceinline__ void glds16_batch(const void* const (&src)[N], const uint32_t (&dst)[N]) {
    static_assert(N == 1 || N == 2 || N == 4 || N == 8, "glds16_batch: unsupported count");
    uint32_t keep;
    if constexpr (N == 1) {
        glds16(src[0], dst[0]);
    } else if constexpr (N == 2) {
        asm volatile("s_mov_b32 %0, m0\n\t"
                     "s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, off\n\t"
                     "s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, off\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep) : "s"(dst[0]), "s"(dst[1]), "v"(src[0]), "v"(src[1]) : "memory");
    } else if constexpr (N == 4) {
        asm volatile("s_mov_b32 %0, m0\n\t"
                     "s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %5, off\n\t"
                     "s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %6, off\n\t"
                     "s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %7, off\n\t"
                     "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %8, off\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep) : "s"(dst[0]), "s"(dst[1]), "s"(dst[2]), "s"(dst[3]),
                       "v"(src[0]), "v"(src[1]), "v"(src[2]), "v"(src[3]) : "memory");
    } else {
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
                     : "=&s"(keep) : "s"(dst[0]), "s"(dst[1]), "s"(dst[2]), "s"(dst[3]), "s"(dst[4]), "s"(dst[5]), "s"(dst[6]), "s"(dst[7]),
                       "v"(src[0]), "v"(src[1]), "v"(src[2]), "v"(src[3]), "v"(src[4]), "v"(src[5]), "v"(src[6]), "v"(src[7]) : "memory");
    }
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" :: "i"(N) : "memory"); }

template <int N, int STEP>
__device__ __forceinline__ void wait_tiles_younger(int younger) {
    // wait until all but `younger` tiles (STEP DMA instructions each) of this wave have landed
    if constexpr (N == 0) { wait_vmcnt<0>(); }
    else { if (younger >= N) wait_vmcnt<(N * STEP > 63 ? 63 : N * STEP)>(); else wait_tiles_younger<N - 1, STEP>(younger); }   // vmcnt is 6 bits: a lower bound only waits longer
}

inline int check_launch() { return hipGetLastError() == hipSuccess ? M355_OK : M355_ERR_LAUNCH; }

// Tuning and A/B switches exist in the DEV build only (make -C csrc dev: -DM355_DEV -> lib/libmi355attn_dev.so, which
// also holds the reference kernels and the timing probes).  The product library reads no environment variable, has one
// path per operation and no mutable state besides per-device lookups of immutable attributes.
#ifdef M355_DEV
inline int dev_env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
inline bool dev_env_set(const char* name) { return getenv(name) != nullptr; }
inline char dev_env_char(const char* name) { const char* e = getenv(name); return e ? e[0] : '\0'; }
#else
inline int dev_env_int(const char*, int dflt) { return dflt; }
inline bool dev_env_set(const char*) { return false; }
inline char dev_env_char(const char*) { return '\0'; }
#endif

// Compute units of the CURRENT device, looked up once per device index (every launch shape follows the device it runs on).
inline int cu_count() {
    constexpr int kMaxDev = 64;
    static std::atomic<int> cache[kMaxDev];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return 256;
    int v = cache[dev].load(std::memory_order_relaxed);
    if (v <= 0) {
        v = 256;
        (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev);
        if (v <= 0) v = 256;
        cache[dev].store(v, std::memory_order_relaxed);
    }
    return v;
}

}  // namespace m355
