// Planned paged decode attention on the matrix cores (gfx950).
//
// Same plan, same partial format and same merge kernel as decode_stream.hip; what changes is the inner product
// engine.  The VALU kernel spends ~200 vector instructions per 16 tokens (dot2 + DPP reductions + per-lane online
// softmax + FMA accumulation) and measured "compute only" it takes as long as streaming the KV bytes does.  Here a
// wave consumes 32-token stages with v_mfma_f32_16x16x32_bf16:
//
//   K, V rows      HBM -> LDS with global_load_lds_dwordx4 (no VGPR round trip), R stages deep, whole 2*D-byte rows
//                  (coalesced), XOR-swizzled by choosing which 16-byte chunk each lane fetches
//   S^T[tok, q]  = K[32 x D] . Q^T[D x 16]     2 x D/32 MFMAs; K fragments are ds_read_b128 rows, Q^T fragments stay in
//                  registers (columns >= G are zero).  A lane then owns ONE q head (column lane&15) and 4+4 tokens
//   softmax        one running max per q head shared by the four 16-lane groups (two permlane swaps), exp2 on 8 values,
//                  the accumulator is rescaled only when some max moved
//   O^T[d, q]   += V^T[D x 32] . P^T[32 x 16]   D/16 MFMAs; P^T is the S^T accumulator re-packed to bf16 in registers
//                  (its k index is the lane group and register, the V^T fragments use the same token order), V^T
//                  fragments are ds_read_b64_tr_b16 transposed reads of the row-major V stage
//
// ~60 vector instructions and 8 MFMAs per 32 tokens instead of ~400.  Token order inside a stage for the k index of
// the second product: lane group kg (lane>>4), element j -> token 4*kg + j (j < 4), 16 + 4*kg + (j-4) (j >= 4).
//
// LDS per wave: R stages of 64*D bytes (K then V), q rows of kQP pieces, the range's slots, the range record.
// All loads inside the loop are LDS-DMA issued and waited for with hand-counted vmcnt (see common.h).
#include <stdlib.h>

#include "stream_plan.h"

#ifndef M355_DECODE_POLICY
#define M355_DECODE_POLICY 1      // K/V are streamed once per step: non-temporal loads (common.h glds16p)
#endif
#define M355_DMA16 glds16p<M355_DECODE_POLICY>
namespace m355 {

constexpr int kQP = 4;                          // pieces whose q rows are staged in LDS at a time
constexpr int kLdsPerCU = 160 * 1024;

__host__ __device__ inline int mfma_lds_bytes(int D, int G, int R, int slots_cap) {
    return R * 64 * D * 2 + kQP * G * 2 * D + slots_cap * 4 + 64 * 4;
}

// ---- In-kernel merge (DEV build only, M355_MERGE_IN_KERNEL=1): the pieces of a cut request count themselves in and the
// last one to arrive merges, instead of the stream_merge_kernel launch.  Built and measured in round 3 and SLOWER on every
// small-launch shape it was meant for ((8,1,128) B=64: 22.2 against 17.7 us; 32 768 tokens x B=1: 20.2 against 14.3;
// profiles/r03_merge_ab.txt): write-through store -> acknowledged -> agent-scope count -> fetch past the own L2 are three
// dependent trips to memory (~2 us each while the chip streams), more than the second launch costs.  Kept as an A/B.
#ifdef M355_DEV
constexpr bool kMergeBuilt = true;
#else
constexpr bool kMergeBuilt = false;
#endif
// partials in the workspace then travel at AGENT scope (sc1: written through by the store, fetched past the reader's own
// L2 by the load): the pieces of one request are streamed on different XCDs, each with an L2 of its own, and the wave that
// merges them runs in the same launch.  Raw buffer accesses carry the policy bit and stay inside hipcc's own wait counting.
constexpr int kAuxAgent = 16;                   // cache policy operand of the buffer builtins on gfx950: sc1
__device__ __forceinline__ __amdgpu_buffer_rsrc_t ws_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
template <int VEC>
__device__ __forceinline__ void ld_agent_vec(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff, float (&o)[VEC]) {
    if constexpr (VEC == 1) {
        o[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, kAuxAgent));
    } else if constexpr (VEC == 2) {
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, kAuxAgent);
        const uint32_t t0 = t[0], t1 = t[1];              // (hipcc 7.2: __builtin_bit_cast of a vector ELEMENT reads element 0)
        o[0] = __builtin_bit_cast(float, t0); o[1] = __builtin_bit_cast(float, t1);
    } else {
#pragma unroll
        for (int x = 0; x < VEC / 4; ++x) {
            const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(r, voff + 16 * x, soff, kAuxAgent);
#pragma unroll
            for (int y = 0; y < 4; ++y) { const uint32_t ty = t[y]; o[4 * x + y] = __builtin_bit_cast(float, ty); }
        }
    }
}
template <int VEC>
__device__ __forceinline__ void st_agent_vec(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff, const float (&o)[VEC]) {
    if constexpr (VEC == 1) {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, o[0]), r, voff, soff, kAuxAgent);
    } else if constexpr (VEC == 2) {
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        __builtin_amdgcn_raw_buffer_store_b64(u32x2{__builtin_bit_cast(uint32_t, o[0]), __builtin_bit_cast(uint32_t, o[1])}, r, voff, soff, kAuxAgent);
    } else {
#pragma unroll
        for (int x = 0; x < VEC / 4; ++x)
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{__builtin_bit_cast(uint32_t, o[4 * x]), __builtin_bit_cast(uint32_t, o[4 * x + 1]),
                                                         __builtin_bit_cast(uint32_t, o[4 * x + 2]), __builtin_bit_cast(uint32_t, o[4 * x + 3])},
                                                   r, voff + 16 * x, soff, kAuxAgent);
    }
}

// One wave combines `count` partials (first, first + stride, ...) of kv head h: a lane owns VEC consecutive values of
// O[G][D] and the (m, l) of their q head.  PB partials' loads are in flight together (one memory latency per batch).
template <int D, int G>
__device__ __forceinline__ void merge_partials(const StreamParams& p, int h, int lane, int first, int count, int stride,
                                               float& gm, float& den, float (&acc)[G * D / 64]) {
    constexpr int VEC = G * D / 64;
    constexpr int PB = VEC >= 16 ? 4 : 8;
    const int e0 = lane * VEC, g = e0 / D;
    const int64_t hq0 = (int64_t)first * p.Hq + h * G;
    const __amdgpu_buffer_rsrc_t ro = ws_rsrc(p.ws_o + hq0 * D);
    const __amdgpu_buffer_rsrc_t rml = ws_rsrc(p.ws_ml + hq0 * 2);
    const uint32_t step_o = (uint32_t)stride * p.Hq * D * 4, step_ml = (uint32_t)stride * p.Hq * 8;
    gm = kNegBig; den = 0.f;
#pragma unroll
    for (int x = 0; x < VEC; ++x) acc[x] = 0.f;
    for (int j0 = 0; j0 < count; j0 += PB) {
        float ml[PB][2], oo[PB][VEC];
#pragma unroll
        for (int j = 0; j < PB; ++j) {
            const uint32_t jj = (uint32_t)__builtin_amdgcn_readfirstlane(j0 + j < count ? j0 + j : count - 1);
            ld_agent_vec<2>(rml, g * 8, jj * step_ml, ml[j]);
            ld_agent_vec<VEC>(ro, e0 * 4, jj * step_o, oo[j]);
        }
        float bm = gm;
#pragma unroll
        for (int j = 0; j < PB; ++j) bm = fmaxf(bm, j0 + j < count ? ml[j][0] : kNegBig);
        const float resc = exp2_fast(gm - bm);
        den *= resc;
#pragma unroll
        for (int x = 0; x < VEC; ++x) acc[x] *= resc;
        gm = bm;
#pragma unroll
        for (int j = 0; j < PB; ++j) {
            const float wgt = j0 + j < count ? exp2_fast(ml[j][0] - gm) : 0.f;
            den = fmaf(wgt, ml[j][1], den);
#pragma unroll
            for (int x = 0; x < VEC; ++x) acc[x] = fmaf(wgt, oo[j][x], acc[x]);
        }
    }
}

template <int VEC, bool F16>
__device__ __forceinline__ void store_out_vec(uint16_t* op, const float (&acc)[VEC], float inv) {
    if constexpr (VEC == 1) {
        *op = (uint16_t)(pack16<F16>(acc[0] * inv, 0.f) & 0xffffu);
    } else {
        uint32_t wds[VEC / 2];
#pragma unroll
        for (int x = 0; x < VEC / 2; ++x) wds[x] = pack16<F16>(acc[2 * x] * inv, acc[2 * x + 1] * inv);
        if constexpr (VEC == 2) *reinterpret_cast<uint32_t*>(op) = wds[0];
        else if constexpr (VEC == 4) *reinterpret_cast<uint2*>(op) = make_uint2(wds[0], wds[1]);
        else {
#pragma unroll
            for (int x = 0; x < VEC / 8; ++x)
                *reinterpret_cast<u32x4*>(op + 8 * x) = u32x4{wds[4 * x], wds[4 * x + 1], wds[4 * x + 2], wds[4 * x + 3]};
        }
    }
}

template <int D, int G, int R, bool STAMP, bool F16 = false>
// (leading scalars: copies of the fields of `p` that address a wave's first record and slots; they are preloaded into SGPRs at
// wave launch - Makefile: -amdgpu-kernarg-preload-count - so those first loads do not wait for a load of the argument block)
__global__ __launch_bounds__(64) void decode_mfma_kernel(const int32_t* plan_hot, const int64_t off_rec_hot, const int64_t off_slots_hot,
                                                         const int32_t Hkv_hot, const int32_t rot_hot, const StreamParams p,
                                                         const int slots_cap) {
    using M = MG<D>;
    static_assert(G <= 16, "q heads of a group sit in the 16 columns of one MFMA");
    extern __shared__ __attribute__((aligned(1024))) uint8_t smem[];
    uint8_t* const ringp = smem;
    uint8_t* const qsp = smem + R * M::STAGEB;
    int32_t* const slots = reinterpret_cast<int32_t*>(qsp + kQP * G * M::ROWB);
    int32_t* const rec = slots + slots_cap;
    int32_t* const pc = rec + 4;                         // piece k: pc[4k .. 4k+3] = {request, start, length, partial index}
    int32_t* const cutp = rec + kRecCut;                 // piece k: {pieces its request is cut into (0: whole), its ordinal}

    sstamp<STAMP>(p, 0);
    const int lane = threadIdx.x;
    const int n = lane & 15;                             // q column / K row / d column, by operand
    const int kg = lane >> 4;                            // 16-lane group = k sub-block
    const int w = blockIdx.x / Hkv_hot;
    // Workgroups go round-robin over the 8 XCDs, so with h = blockIdx % Hkv one XCD would stream one head only: one
    // 2*D-byte column of every Hkv*2*D-byte token row, i.e. a fixed value of the low address bits its L2 channels are
    // interleaved on.  Rotating the head every p.rot ranges lets every XCD touch every column (measured 35.3 -> 31.5 us).
    const int h = (blockIdx.x + (rot_hot > 0 ? w / rot_hot : 0)) % Hkv_hot;
    const int32_t* plan = plan_hot;
    const uint8_t* kbase = reinterpret_cast<const uint8_t*>(p.k_pool) + (int64_t)h * M::ROWB;
    const uint8_t* vbase = reinterpret_cast<const uint8_t*>(p.v_pool) + (int64_t)h * M::ROWB;
    const int64_t row_bytes = p.row_elems * 2;
    const uint32_t ring0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)ringp);

    // per-lane constants
    const int drow = lane / M::C, dch = lane % M::C;     // DMA: row inside the instruction, chunk position
    int koff[M::NI], voff[M::NI];
#pragma unroll
    for (int i = 0; i < M::NI; ++i) {
        const int row = i * M::RPI + drow;
        koff[i] = 16 * (dch ^ M::sK(row));
        voff[i] = 16 * (dch ^ M::sV(row));
    }
    const int ka0 = n * M::ROWB + 16 * (kg ^ M::sK(n));  // K fragment (row n [+16], k-step s): ka0 ^ 64 s
    const int vrow = 4 * kg + (n >> 2);
    const int va0 = vrow * M::ROWB + 16 * (M::sV(vrow) | ((n & 3) >> 1)) + 8 * (n & 1);   // V^T block (d-tile dt): va0 ^ 32 dt
    const int qa0 = (n < G ? n : 0) * M::ROWB + 16 * kg;                                  // Q^T fragment (k-step s): + 64 s

    f32x4 O[M::DT];
    float m_run = kNegBig, l_run = 0.f;
    bf16x8 qf[M::KS];

    const bool fused = p.k_new != nullptr;
    uint32_t endmask = 0;                                // bit k: piece k of the range ends its request
    // partial pieces of the current range that still have to count themselves in (a range's first and last piece only)
    int nd = 0, da_b = 0, da_pidx = 0, da_tot = 0, da_k = 0, dz_b = 0, dz_pidx = 0, dz_tot = 0, dz_k = 0;
    const __amdgpu_buffer_rsrc_t ws_o_r = ws_rsrc(p.ws_o), ws_ml_r = ws_rsrc(p.ws_ml);

    auto run_pass = [&](const int p0, const int np) {
        int lk = 0, lit = 0, ck = 0, cit = 0, issued = 0, consumed = 0;
        int l_start = __builtin_amdgcn_readfirstlane(pc[p0 * 4 + 1]);
        int l_len = __builtin_amdgcn_readfirstlane(pc[p0 * 4 + 2]);
        int c_len = l_len;
        // fused KV store: the newest token of a request is read from the step's k/v rows, not from the pool
        bool l_new = false;
        const uint8_t* l_knew = nullptr;
        const uint8_t* l_vnew = nullptr;
        auto new_rows = [&]() {
            l_new = fused && ((endmask >> (p0 + lk)) & 1u);
            if (l_new) {
                const int64_t bq = __builtin_amdgcn_readfirstlane(pc[(p0 + lk) * 4]);
                l_knew = reinterpret_cast<const uint8_t*>(p.k_new + bq * p.kn_stride) + h * M::ROWB;
                l_vnew = reinterpret_cast<const uint8_t*>(p.v_new + bq * p.vn_stride) + h * M::ROWB;
            }
        };
        new_rows();

        auto issue = [&](const int RS) {                  // RS: ring slot, a constant after unrolling
            const uint32_t sbase = ring0 + RS * M::STAGEB;
            // all slot look-ups first (one LDS round trip; the DMA asm statements are memory barriers for the compiler)
            int64_t ro[M::NI];
            int tt[M::NI];
#pragma unroll
            for (int i = 0; i < M::NI; ++i) {
                tt[i] = lit * 32 + i * M::RPI + drow;
                tt[i] = tt[i] < l_len ? tt[i] : l_len - 1;   // clamp: rows past the piece are masked in the scores
                ro[i] = (int64_t)slots[l_start + tt[i]] * row_bytes;
            }
            if (l_new && (lit + 1) * 32 >= l_len) {       // wave-uniform: this stage holds the request's newest token
#pragma unroll
                for (int i = 0; i < M::NI; ++i) {
                    const bool fresh = tt[i] == l_len - 1;
                    M355_DMA16((fresh ? l_knew : kbase + ro[i]) + koff[i], sbase + 1024 * i);
                    M355_DMA16((fresh ? l_vnew : vbase + ro[i]) + voff[i], sbase + M::TILEB + 1024 * i);
                }
            } else {
#pragma unroll
                for (int i = 0; i < M::NI; ++i) {
                    M355_DMA16(kbase + ro[i] + koff[i], sbase + 1024 * i);
                    M355_DMA16(vbase + ro[i] + voff[i], sbase + M::TILEB + 1024 * i);
                }
            }
            ++issued;
            ++lit;
            if (lit * 32 >= l_len) {
                ++lk; lit = 0;
                if (lk < np) {
                    l_start = __builtin_amdgcn_readfirstlane(pc[(p0 + lk) * 4 + 1]);
                    l_len = __builtin_amdgcn_readfirstlane(pc[(p0 + lk) * 4 + 2]);
                    new_rows();
                }
            }
        };
        auto begin_piece = [&]() {
#pragma unroll
            for (int s = 0; s < M::KS; ++s) {
                u32x4 t = *reinterpret_cast<const u32x4*>(qsp + ck * (G * M::ROWB) + qa0 + 64 * s);
                if (n >= G) t = u32x4{0u, 0u, 0u, 0u};
                qf[s] = __builtin_bit_cast(bf16x8, t);
            }
            m_run = kNegBig; l_run = 0.f;
#pragma unroll
            for (int dt = 0; dt < M::DT; ++dt) O[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        };
        // A finished piece's results wait in registers until the NEXT stage has been waited for: stored at once they
        // would sit in front of that stage in the vmcnt queue and the counted wait would include their write acknowledge.
        bool pend = false;
        int pend_bq = 0, pend_pidx = 0, pend_tot = 0, pend_k = 0;
        float pend_m = 0.f, pend_l = 0.f;
        f32x4 pend_O[M::DT];
        auto flush_piece = [&]() {
            pend = false;
            const int bq = pend_bq, pidx = pend_pidx;
            const float l = pend_l, m_run = pend_m;
            const f32x4 (&O)[M::DT] = pend_O;
            if (n < G) {
                if (pidx < 0) {                          // the piece is the whole request: final output
                    const float inv = 1.f / l;
                    uint16_t* op = p.out + (int64_t)bq * p.o_stride + (int64_t)(h * G + n) * D + 4 * kg;
#pragma unroll
                    for (int dt = 0; dt < M::DT; ++dt) {
                        uint32_t w0 = pack16<F16>(O[dt][0] * inv, O[dt][1] * inv), w1 = pack16<F16>(O[dt][2] * inv, O[dt][3] * inv);
                        *reinterpret_cast<uint2*>(op + 16 * dt) = make_uint2(w0, w1);
                    }
                } else if (kMergeBuilt && p.merge_in_kernel) {
                    const uint32_t idx = (uint32_t)pidx * p.Hq + h * G + n;
#pragma unroll
                    for (int dt = 0; dt < M::DT; ++dt) {
                        const float o4[4] = {O[dt][0], O[dt][1], O[dt][2], O[dt][3]};
                        st_agent_vec<4>(ws_o_r, (idx * D + 4 * kg + 16 * dt) * 4, 0, o4);
                    }
                    if (kg == 0) { const float ml2[2] = {m_run, l}; st_agent_vec<2>(ws_ml_r, idx * 8, 0, ml2); }
                } else {
                    const int64_t idx = (int64_t)pidx * p.Hq + h * G + n;
                    float* wo = p.ws_o + idx * D + 4 * kg;
#pragma unroll
                    for (int dt = 0; dt < M::DT; ++dt) *reinterpret_cast<f32x4*>(wo + 16 * dt) = O[dt];
                    if (kg == 0) { p.ws_ml[idx * 2] = m_run; p.ws_ml[idx * 2 + 1] = l; }
                }
            }
            if (kMergeBuilt && pidx >= 0 && p.merge_in_kernel) {        // (wave-uniform) the piece counts itself in when the range is done
                if (nd == 0) { da_b = bq; da_pidx = pidx; da_tot = pend_tot; da_k = pend_k; }
                else { dz_b = bq; dz_pidx = pidx; dz_tot = pend_tot; dz_k = pend_k; }
                ++nd;
            }
        };
        auto end_piece = [&]() {
            if (pend) flush_piece();                     // (two piece ends without a wait in between: tiny pieces)
            float a, b;
            lane_pair<16>(l_run, a, b);
            float l = a + b;
            lane_pair<32>(l, a, b);
            pend_l = a + b;
            pend_m = m_run;
            pend_bq = __builtin_amdgcn_readfirstlane(pc[(p0 + ck) * 4 + 0]);
            pend_pidx = __builtin_amdgcn_readfirstlane(pc[(p0 + ck) * 4 + 3]);
            pend_tot = __builtin_amdgcn_readfirstlane(cutp[(p0 + ck) * 2]);
            pend_k = __builtin_amdgcn_readfirstlane(cutp[(p0 + ck) * 2 + 1]);
#pragma unroll
            for (int dt = 0; dt < M::DT; ++dt) pend_O[dt] = O[dt];
            pend = true;
        };
        auto consume = [&](const int RS) {
            const uint8_t* st = ringp + RS * M::STAGEB;
            if (cit == 0) begin_piece();
            f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < M::KS; ++s) {
                const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(st + (ka0 ^ (64 * s)));
                const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(st + 16 * M::ROWB + (ka0 ^ (64 * s)));
                s0 = mfma_16x16x32<F16>(a0, qf[s], s0);
                s1 = mfma_16x16x32<F16>(a1, qf[s], s1);
            }
            const int rem = c_len - cit * 32;            // valid tokens of this stage
            if (rem < 32) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (4 * kg + r >= rem) s0[r] = kNegBig;
                    if (16 + 4 * kg + r >= rem) s1[r] = kNegBig;
                }
            }
            float mx = fmaxf(fmaxf(fmaxf(s0[0], s0[1]), fmaxf(s0[2], s0[3])), fmaxf(fmaxf(s1[0], s1[1]), fmaxf(s1[2], s1[3])));
            float a, b;
            lane_pair<16>(mx, a, b);
            mx = fmaxf(a, b);
            lane_pair<32>(mx, a, b);
            mx = fmaxf(a, b);
            // lazy running max: it moves only when beaten by more than 2^kLazyMax, so rescaling the O^T accumulators is
            // rare; p <= 2^kLazyMax stays far inside bf16 / fp32 range and the partial (m, l, o) stay consistent
            const float m_tile = mx * p.scale_log2;
            const float m_new = m_tile > m_run + kLazyMax ? m_tile : m_run;
            const float alpha = exp2_fast(m_run - m_new);
            m_run = m_new;
            float sum = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s0[r] = exp2_fast(fmaf(s0[r], p.scale_log2, -m_new));
                s1[r] = exp2_fast(fmaf(s1[r], p.scale_log2, -m_new));
                sum += s0[r] + s1[r];
            }
            l_run = fmaf(l_run, alpha, sum);
            if (__any(alpha != 1.f)) {
#pragma unroll
                for (int dt = 0; dt < M::DT; ++dt) O[dt] *= alpha;
            }
            const u32x4 pw = {pack16<F16>(s0[0], s0[1]), pack16<F16>(s0[2], s0[3]), pack16<F16>(s1[0], s1[1]), pack16<F16>(s1[2], s1[3])};
            const bf16x8 pf = __builtin_bit_cast(bf16x8, pw);
#pragma unroll
            for (int dt = 0; dt < M::DT; ++dt) {
                const uint8_t* va = st + M::TILEB + (va0 ^ (32 * dt));
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (s16x4 __attribute__((address_space(3)))*)(uintptr_t)(uint32_t)(uintptr_t)va);
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (s16x4 __attribute__((address_space(3)))*)(uintptr_t)(uint32_t)(uintptr_t)(va + 16 * M::ROWB));
                const s16x8 av = s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                O[dt] = mfma_16x16x32<F16>(__builtin_bit_cast(bf16x8, av), pf, O[dt]);
            }
            ++consumed;
            ++cit;
            if (cit * 32 >= c_len) {
                end_piece();
                ++ck; cit = 0;
                if (ck < np) c_len = __builtin_amdgcn_readfirstlane(pc[(p0 + ck) * 4 + 2]);
            }
        };

        wait_vmcnt<0>();                                 // stores of the previous pass: keep the DMA count exact
        sstamp<STAMP>(p, 2);
        // ---- fill the ring, then: refill the slot consumed last, wait for the oldest stage, consume it
#pragma unroll
        for (int j = 0; j < R - 1; ++j)
            if (lk < np) issue(j);
        // ---- q rows of the pass's pieces -> LDS, behind the first stages' DMA (ordinary loads: the compiler's wait
        // for them also covers the older DMA, which the first consume needs anyway)
        for (int i = lane; i < np * G * M::C; i += kWave) {
            const int k = i / (G * M::C), c = i - k * (G * M::C);
            *reinterpret_cast<u32x4*>(qsp + i * 16) =
                *reinterpret_cast<const u32x4*>(p.q + (int64_t)pc[(p0 + k) * 4] * p.q_stride + (int64_t)h * G * D + c * 8);
        }
        __syncthreads();
        sstamp<STAMP>(p, 3);
        while (ck < np) {
#pragma unroll
            for (int j = 0; j < R; ++j) {
                if (ck < np) {
                    if (lk < np) issue((j + R - 1) % R);
                    wait_tiles_younger<R - 1, 2 * M::NI>(issued - consumed - 1);
                    if (pend) flush_piece();
                    consume(j);
                }
            }
        }
        if (pend) flush_piece();
        if (fused) {
            // the new rows of this head -> their pool slots, AFTER the stream: a store in front of it sits in the
            // same vmcnt queue as the DMA stages, and the first counted wait behind it would also wait for its
            // write acknowledge
            for (int k = 0; k < np; ++k) {
                if (!((endmask >> (p0 + k)) & 1u)) continue;
                if (lane < 2 * M::C) {
                    const int c = lane % M::C;
                    const int64_t bq = pc[(p0 + k) * 4];
                    const int64_t slot = slots[pc[(p0 + k) * 4 + 1] + pc[(p0 + k) * 4 + 2] - 1];
                    const uint16_t* src = (lane < M::C ? p.k_new + bq * p.kn_stride : p.v_new + bq * p.vn_stride) + h * D + c * 8;
                    uint16_t* dst = const_cast<uint16_t*>(lane < M::C ? p.k_pool : p.v_pool) + slot * p.row_elems + h * D + c * 8;
                    *reinterpret_cast<u32x4*>(dst) = *reinterpret_cast<const u32x4*>(src);
                }
            }
        }
        __syncthreads();                                 // rec / q rows / slots are rewritten next
    };

    for (int r = w; r < p.max_ranges; r += p.nwh) {
        // ---- the record and (speculatively) the first slots in one round of loads
        const int32_t* grec = plan + off_rec_hot + (int64_t)r * kRec;
        const int32_t* gslots = plan + off_slots_hot + (int64_t)r * kRMax;
        int32_t rv = 0, sv[kSpec];
        if (lane < kRec) rv = grec[lane];
#pragma unroll
        for (int k = 0; k < kSpec; ++k) sv[k] = gslots[lane + kWave * k];
        if (lane < kRec) rec[lane] = rv;
#pragma unroll
        for (int k = 0; k < kSpec; ++k)
            if (lane + kWave * k < slots_cap) slots[lane + kWave * k] = sv[k];
        __syncthreads();
        const int32_t nt = __builtin_amdgcn_readfirstlane(rec[0]);
        const int32_t np_rec = __builtin_amdgcn_readfirstlane(rec[1]);
        if (nt == 0) break;                              // past the last range
        if (nt > kSpec * kWave) {
            for (int i = kSpec * kWave + lane; i < nt && i < slots_cap; i += kWave) slots[i] = gslots[i];
            __syncthreads();
        }
        sstamp<STAMP>(p, 1);
        // pieces come from the record, or - more than kMaxPieces requests in the range - from scanning cum in rounds
        const bool scanning = np_rec < 0;
        const int32_t* cum = plan + p.off_cum;
        const int32_t* pbase = plan + p.off_pbase;
        const int32_t g0 = __builtin_amdgcn_readfirstlane(rec[3]);
        const int32_t Rt = plan[1];
        int b_next = __builtin_amdgcn_readfirstlane(rec[2]);
        int32_t done = 0;
        int np = np_rec;
        endmask = __builtin_amdgcn_readfirstlane(rec[kRecEnds]);
        do {
            if (scanning) {
                __syncthreads();
                const int bb = b_next + lane;
                int32_t cs = 0, ce = 0;
                if (lane < kMaxPieces && bb < p.B) { cs = cum[bb]; ce = cum[bb + 1]; }
                const int32_t lo = max(cs, g0 + done), hi = min(ce, g0 + nt);
                const bool have = lane < kMaxPieces && bb < p.B && hi > lo;
                const unsigned long long mask = __ballot(have);
                np = __popcll(mask);
                if (have) {
                    const int k = __popcll(mask & ((1ull << lane) - 1));
                    pc[k * 4 + 0] = bb;
                    pc[k * 4 + 1] = lo - g0;
                    pc[k * 4 + 2] = hi - lo;
                    const bool whole = lo == cs && hi == ce;
                    pc[k * 4 + 3] = whole ? -1 : pbase[bb] + (r - cs / Rt);
                    cutp[k * 2] = whole ? 0 : (ce - 1) / Rt - cs / Rt + 1;
                    cutp[k * 2 + 1] = whole ? 0 : r - cs / Rt;
                }
                const unsigned long long ends = __ballot(have && hi == ce && plan[p.off_full + bb] != 0);
                endmask = 0;
                for (int j = 0, k = 0; j < kMaxPieces; ++j)
                    if ((mask >> j) & 1ull) { endmask |= (uint32_t)((ends >> j) & 1ull) << k; ++k; }
                const int last = 63 - __clzll(mask | 1ull);
                done = __builtin_amdgcn_readfirstlane((np > 0 ? __shfl(hi, last) : g0 + nt) - g0);
                b_next = __builtin_amdgcn_readfirstlane(np > 0 ? b_next + last + 1 : b_next + kMaxPieces);
                __syncthreads();
            }
            for (int p0 = 0; p0 < np; p0 += kQP) run_pass(p0, min(kQP, np - p0));
        } while (scanning && done < nt);
        // ---- the range's partial pieces count themselves in; whoever completes a count merges
        if (kMergeBuilt && nd > 0) {
            wait_vmcnt<0>();                             // this wave's partials have been written through
            constexpr int VEC = G * D / 64;
            const int e0 = lane * VEC, gq = e0 / D, d0 = e0 - gq * D;
            int32_t* const arrivals = const_cast<int32_t*>(plan) + p.off_arrivals;
            if (nd > 2 && lane == 0)                     // (cannot happen: only a range's first and last piece are parts of longer requests)
                __hip_atomic_store(const_cast<int32_t*>(plan) + M355_PLAN_ERROR_WORD, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int i = 0; i < (nd < 2 ? nd : 2); ++i) {
                const int bq = i ? dz_b : da_b, pidx = i ? dz_pidx : da_pidx, tot = i ? dz_tot : da_tot, k = i ? dz_k : da_k;
                const int gs = merge_group(tot);
                const int gi = k / gs, ng = (tot + gs - 1) / gs;
                const int n_g = tot - gi * gs < gs ? tot - gi * gs : gs;
                const int pf = pidx - k + gi * gs;       // the group's first partial
                int32_t* const c1 = arrivals + (int64_t)pf * p.Hkv + h;
                int old = 0;
                if (lane == 0) old = __hip_atomic_fetch_add(c1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                old = __builtin_amdgcn_readfirstlane(old);
                if (old != n_g - 1) continue;
                if (lane == 0) __hip_atomic_store(c1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
                float gm, den, acc[VEC];
                merge_partials<D, G>(p, h, lane, pf, n_g, 1, gm, den, acc);
                if (ng > 1) {                            // two levels: leave ONE partial in the place of the group's first piece
                    const uint32_t idx = (uint32_t)pf * p.Hq + h * G + gq;
                    st_agent_vec<VEC>(ws_o_r, (idx * D + d0) * 4, 0, acc);
                    if (d0 == 0) { const float ml2[2] = {gm, den}; st_agent_vec<2>(ws_ml_r, idx * 8, 0, ml2); }
                    wait_vmcnt<0>();
                    int32_t* const c2 = arrivals + (int64_t)(pidx - k + 1) * p.Hkv + h;   // (never a group's first partial: groups are >= 16 apart)
                    old = 0;
                    if (lane == 0) old = __hip_atomic_fetch_add(c2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    old = __builtin_amdgcn_readfirstlane(old);
                    if (old != ng - 1) continue;
                    if (lane == 0) __hip_atomic_store(c2, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    merge_partials<D, G>(p, h, lane, pidx - k, ng, gs, gm, den, acc);
                }
                store_out_vec<VEC, F16>(p.out + (int64_t)bq * p.o_stride + (int64_t)(h * G + gq) * D + d0, acc, 1.f / den);
            }
            nd = 0;
        }
    }
    if constexpr (STAMP) { sstamp<STAMP>(p, 4); wait_vmcnt<0>(); sstamp<STAMP>(p, 5); }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
// Ring depth 2, up to 8 waves per CU, ranges of >= 128 tokens (4 stages; 64 and 32 measured slower); the dev build can vary them.
static int mfma_ring() {
    static const int v = [] { const int x = dev_env_int("M355_MFMA_RING", 2); return x < 2 ? 2 : (x > 6 ? 6 : x); }();
    return v;
}
static int mfma_max_wpc() {
    static const int v = [] { const int x = dev_env_int("M355_MFMA_WPC", 8); return x < 1 ? 1 : (x > 8 ? 8 : x); }();
    return v;
}
int mfma_rmin() {
    static const int v = [] { const int x = dev_env_int("M355_MFMA_RMIN", 128); return x < 32 ? 32 : x; }();
    return v;
}

// slots a wave stages = the longest range the plan can make for this capacity
static int mfma_slots_cap(int nwh, int64_t cap) {
    int64_t rt = (cap + nwh - 1) / nwh;
    if (rt < mfma_rmin()) rt = mfma_rmin();
    if (rt > kRMax) rt = kRMax;
    return (int)((rt + 63) / 64 * 64);
}

int mfma_waves_per_cu(int Hkv, int D, int G, int64_t cap) {
    const int R = mfma_ring();
    for (int wpc = mfma_max_wpc(); wpc > 1; --wpc) {
        int nwh = cu_count() * wpc / Hkv;
        if (nwh < 1) nwh = 1;
        if ((int64_t)mfma_lds_bytes(D, G, R, mfma_slots_cap(nwh, cap)) * wpc <= kLdsPerCU) return wpc;
    }
    return 1;
}

template <int D, int G>
static int launch_mfma(const StreamParams& p, int64_t cap, hipStream_t st, bool stamp, bool f16) {
    const int R = f16 ? 2 : mfma_ring();               // (the tuning rings and the stamp build exist for bf16 only)
    const int slots_cap = mfma_slots_cap(p.nwh, cap);
    const int lds = mfma_lds_bytes(D, G, R, slots_cap);
    if (lds > 64 * 1024) return M355_ERR_UNSUPPORTED;
    const dim3 grid((unsigned)(p.Hkv * p.nwh)), block(64);
#define M355_HOT p.plan, p.off_rec, p.off_slots, p.Hkv, p.rot
    if (f16) {
        hipLaunchKernelGGL((decode_mfma_kernel<D, G, 2, false, true>), grid, block, lds, st, M355_HOT, p, slots_cap);
        return M355_OK;
    }
#ifdef M355_DEV
#define M355_RING(RR) if (R == RR) { \
        if (stamp) hipLaunchKernelGGL((decode_mfma_kernel<D, G, RR, true>), grid, block, lds, st, M355_HOT, p, slots_cap); \
        else hipLaunchKernelGGL((decode_mfma_kernel<D, G, RR, false>), grid, block, lds, st, M355_HOT, p, slots_cap); }
    M355_RING(2) M355_RING(3) M355_RING(4) M355_RING(5) M355_RING(6)
#undef M355_RING
#else
    (void)stamp;
    hipLaunchKernelGGL((decode_mfma_kernel<D, G, 2, false>), grid, block, lds, st, M355_HOT, p, slots_cap);
#endif
    return M355_OK;
}

#undef M355_HOT

int launch_stream_mfma(const StreamParams& p, int D, int G, int64_t cap, hipStream_t st, bool stamp, bool f16) {
#define M355_CASE(DD, GG) if (D == DD && G == GG) return launch_mfma<DD, GG>(p, cap, st, stamp, f16);
    M355_CASE(64, 1) M355_CASE(64, 2) M355_CASE(64, 4) M355_CASE(64, 8)
    M355_CASE(128, 1) M355_CASE(128, 2) M355_CASE(128, 4) M355_CASE(128, 8)
#undef M355_CASE
    return M355_ERR_UNSUPPORTED;
}

}  // namespace m355
