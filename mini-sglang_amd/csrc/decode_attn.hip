// Paged decode attention for gfx950 (reference torch_native_backend.py:121-193, 237-276).
//
// HBM-bound: every cached K and V row of every sequence is read exactly once.  Algorithmic bytes per
// launch = 4*Hkv*D*sum(L_i) (bf16 K+V) + 4*B*Hq*D (q read, o write) + 4*sum(ceil(L_i/ps)) + 16*B.
//
// Work decomposition: one 64-lane wave per (query row, kv head, chunk of `chunk` tokens).  A lane owns
// 8 consecutive head-dim elements (one 16-byte load) of a token row, so LPH = D/8 lanes cover one row of
// one kv head and a wave-wide global_load_dwordx4 fetches NT = 64/LPH whole rows (128 B or 256 B each,
// full cache lines).  The G = Hq/Hkv query heads of the group share every K/V load.  q.k uses
// v_dot2c_f32_bf16 on the packed bf16 pairs, the LPH partial sums are combined with DPP adds, softmax is
// the online (running max / running sum) form in fp32 with exp2, and p.v accumulates in fp32 registers.
// Page-table entries of the chunk are staged in LDS once; K/V rows go straight to VGPRs (they are used
// once, an LDS round trip would only add traffic), double-buffered one iteration ahead so that each wave
// keeps 2*U KiB of loads in flight.  Chunks of one sequence are merged by decode_merge_kernel from fp32
// (m, l, acc) partials; sequences that fit one chunk write bf16 output directly.
#ifdef M355_DEV      // the whole file: a dev-build A/B reference (and the row-parallel prefill of M355_PREFILL_MODE=1)
#include <stdlib.h>

#include "common.h"
#include "paged_rows.h"

namespace m355 {

struct DecodeParams {
    const uint16_t* q;
    const uint16_t* k_pool;
    const uint16_t* v_pool;
    const int32_t* page_table;
    const int64_t* row_ids;    // decode: [NQ] page-table row of each query; extend: [B] row of each request
    const int64_t* kv_lens;    // decode: [NQ] tokens each query attends; extend: unused
    const int32_t* cu_q;       // extend mode only (else null): [B+1] first packed query row of each request
    const int64_t* prefix_lens;  // extend mode only: [B]
    uint16_t* out;
    float* ws_o;               // [NQ, Hq, max_splits, D]
    float* ws_ml;              // [NQ, Hq, max_splits, 2]
    int64_t q_stride, o_stride, pt_stride;
    int64_t row_elems;         // Hkv * D: distance between consecutive slots in a pool
    int32_t NQ, B, Hq, Hkv, ps, ps_shift;
    int32_t chunk, max_splits;
    float scale_log2;          // softmax scale * log2(e)
};

// Dev-only timing probe: when set (m355dbg_set_stamp_buffer) and the stamping variant is selected, every wave
// records s_memrealtime (100 MHz) at its phase boundaries: [bid*8 + k].
__device__ unsigned long long* g_stamp_buf = nullptr;
__device__ __forceinline__ void stamp(int k) {
    if (g_stamp_buf && threadIdx.x == 0) g_stamp_buf[(size_t)blockIdx.x * 8 + k] = __builtin_amdgcn_s_memrealtime();
}

constexpr int kMaxChunk = kRowsMaxChunk;   // LDS page staging is sized for page_size 1

// Which page-table row a query reads and how many cached tokens it attends.
// decode: query qi is request qi and sees all kv_lens[qi] tokens (torch_native_backend.py:160-174);
// extend: packed query qi = cu_q[b] + j sits at position prefix_lens[b] + j and sees tokens 0..that (:71-117).
__device__ __forceinline__ void query_row_len(const DecodeParams& p, int qi, int64_t& row, int64_t& L) {
    if (p.cu_q) {
        int lo = 0, hi = p.B - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (p.cu_q[mid] <= qi) lo = mid; else hi = mid - 1;
        }
        row = p.row_ids[lo];
        L = p.prefix_lens[lo] + (qi - p.cu_q[lo]) + 1;
    } else {
        row = p.row_ids[qi];
        L = p.kv_lens[qi];
    }
}


// ------------------------------------------------------------------------------------------------
// pieces shared by the two split kernels
// ------------------------------------------------------------------------------------------------
template <int D, int G>
struct WaveState {
    uint32_t q[G][4];       // packed bf16 pairs: this lane's 8 head-dim elements of each q head of the group
    float m[G], l[G];       // running max (scaled, log2 domain) and running sum
    float o[G][8];          // running output (unnormalised)
};

// Slot of every token of the chunk -> LDS (int32; S < 2^31).  Entries past ntok repeat the last valid slot so
// the main loop never clamps; masking is done on the score.
template <bool POW2>
__device__ __forceinline__ void stage_slots(const DecodeParams& p, int32_t* slots, int lane, int64_t row, int64_t t0,
                                            int ntok, int ntok_padded) {
    const int32_t* pt_row = p.page_table + row * p.pt_stride;
    for (int t = lane; t < ntok_padded; t += kWave) {
        const int64_t tok = t0 + (t < ntok ? t : ntok - 1);
        const int64_t pg = POW2 ? (tok >> p.ps_shift) : (tok / p.ps);
        const int32_t in_pg = (int32_t)(tok - (POW2 ? (pg << p.ps_shift) : pg * p.ps));
        slots[t] = pt_row[pg] * p.ps + in_pg;
    }
}

template <int D, int G>
__device__ __forceinline__ void init_state(const DecodeParams& p, WaveState<D, G>& st, int qi, int h, int dchunk) {
    const uint16_t* qp = p.q + (int64_t)qi * p.q_stride + (int64_t)h * G * D + dchunk * 8;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const u32x4 t = *reinterpret_cast<const u32x4*>(qp + g * D);
        st.q[g][0] = t[0]; st.q[g][1] = t[1]; st.q[g][2] = t[2]; st.q[g][3] = t[3];
        st.m[g] = kNegBig; st.l[g] = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) st.o[g][j] = 0.f;
    }
}

// One tile = U token rows per lane group.  bias[u] is 0 for a real token and kNegBig for a padded one: the
// score becomes ~-1e30, its weight exp2(s - max) underflows to exactly 0 as soon as the group has seen one real
// token; a group that has seen none keeps max = -1e30 and is wiped when groups/chunks are merged.
template <int D, int G, int U>
__device__ __forceinline__ void compute_tile(WaveState<D, G>& st, const u32x4 (&kt)[U], const u32x4 (&vt)[U],
                                             const float (&bias)[U], float scale_log2) {
    constexpr int LPH = D / 8;
    float s[U][G];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float a = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) a = dot2_bf16(st.q[g][w], kt[u][w], a);
            s[u][g] = a;
        }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int g = 0; g < G; ++g) s[u][g] = fmaf(group_sum<LPH>(s[u][g]), scale_log2, bias[u]);

    float alpha[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        float mx = st.m[g];
#pragma unroll
        for (int u = 0; u < U; ++u) mx = fmaxf(mx, s[u][g]);
        alpha[g] = exp2_fast(st.m[g] - mx);
        st.m[g] = mx;
        float sum = 0.f;
#pragma unroll
        for (int u = 0; u < U; ++u) { s[u][g] = exp2_fast(s[u][g] - mx); sum += s[u][g]; }
        st.l[g] = fmaf(st.l[g], alpha[g], sum);
    }
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int j = 0; j < 8; ++j) st.o[g][j] *= alpha[g];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        float vf[8];
#pragma unroll
        for (int w = 0; w < 4; ++w) { vf[2 * w] = bf16_lo(vt[u][w]); vf[2 * w + 1] = bf16_hi(vt[u][w]); }
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int j = 0; j < 8; ++j) st.o[g][j] = fmaf(s[u][g], vf[j], st.o[g][j]);
    }
}

// Merge the NT lane-group states of the wave, then write bf16 output (single-chunk sequence) or the fp32 partial.
template <int D, int G>
__device__ __forceinline__ void finish_wave(const DecodeParams& p, WaveState<D, G>& st, int qi, int h, int split,
                                            int64_t L, int lane) {
    constexpr int LPH = D / 8;
    const int dchunk = lane % LPH;
#pragma unroll
    for (int mask = LPH; mask < kWave; mask <<= 1) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const float m2 = __shfl_xor(st.m[g], mask);
            const float l2 = __shfl_xor(st.l[g], mask);
            const float mx = fmaxf(st.m[g], m2);
            const float wa = exp2_fast(st.m[g] - mx), wb = exp2_fast(m2 - mx);
            st.l[g] = st.l[g] * wa + l2 * wb;
#pragma unroll
            for (int j = 0; j < 8; ++j) st.o[g][j] = st.o[g][j] * wa + __shfl_xor(st.o[g][j], mask) * wb;
            st.m[g] = mx;
        }
    }
    if (lane >= LPH) return;
    const int nsplit = (int)((L + p.chunk - 1) / p.chunk);
    if (nsplit == 1) {
        uint16_t* op = p.out + (int64_t)qi * p.o_stride + (int64_t)h * G * D + dchunk * 8;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const float inv = 1.f / st.l[g];
            u32x4 w;
#pragma unroll
            for (int x = 0; x < 4; ++x) w[x] = pack_bf16(st.o[g][2 * x] * inv, st.o[g][2 * x + 1] * inv);
            *reinterpret_cast<u32x4*>(op + g * D) = w;
        }
    } else {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int64_t idx = ((int64_t)qi * p.Hq + h * G + g) * p.max_splits + split;
            float* wo = p.ws_o + idx * D + dchunk * 8;
            *reinterpret_cast<f32x4*>(wo) = f32x4{st.o[g][0], st.o[g][1], st.o[g][2], st.o[g][3]};
            *reinterpret_cast<f32x4*>(wo + 4) = f32x4{st.o[g][4], st.o[g][5], st.o[g][6], st.o[g][7]};
            if (dchunk == 0) { p.ws_ml[idx * 2] = st.m[g]; p.ws_ml[idx * 2 + 1] = st.l[g]; }
        }
    }
}

#define M355_WORK_ITEM_PROLOGUE()                                                              \
    constexpr int LPH = D / 8;                                                                 \
    constexpr int NT = kWave / LPH;                                                            \
    constexpr int TPI = NT * U;                                                                \
    const int lane = threadIdx.x;                                                              \
    const int dchunk = lane % LPH;                                                             \
    const int tsub = lane / LPH;                                                               \
    uint32_t bid = blockIdx.x;                                                                 \
    const int split = bid % p.max_splits;  bid /= p.max_splits;                                \
    const int h = bid % p.Hkv;                                                                 \
    const int qi = bid / p.Hkv;                                                                \
    int64_t L, row;                                                                            \
    query_row_len(p, qi, row, L);                                                              \
    const int64_t t0 = (int64_t)split * p.chunk;                                               \
    if (t0 >= L) return;                                                                       \
    const int ntok = __builtin_amdgcn_readfirstlane((int)((L - t0 < p.chunk) ? (L - t0) : p.chunk)); \
    const int n_it = (ntok + TPI - 1) / TPI;

// ------------------------------------------------------------------------------------------------
// Variant R: K/V rows straight to VGPRs through NB rotating register tiles: NB-1 tiles ((NB-1) * 2U KiB per
// wave) are in flight while one is computed.  Decode is latency-bound per wave (HBM latency under load is
// microseconds), so the bytes in flight per CU - waves x (NB-1) x 2U KiB - set the achieved bandwidth.
// ------------------------------------------------------------------------------------------------
template <int D, int G, int U, int NB, int W, bool POW2, int MODE = 0>
__global__ __launch_bounds__(64, W) void decode_split_kernel(const DecodeParams p) {
    __shared__ int32_t slots[kMaxChunk + 64];
    if constexpr (MODE == 6) stamp(0);
    M355_WORK_ITEM_PROLOGUE()
    if constexpr (MODE == 6) stamp(1);
    if constexpr (MODE == 4) { if (lane == 0 && ntok == 12345678) p.ws_ml[0] = 1.f; return; }   // timing probe
    stage_slots<POW2>(p, slots, lane, row, t0, ntok, n_it * TPI);
    __syncthreads();

    WaveState<D, G> st;
    init_state<D, G>(p, st, qi, h, dchunk);
    const int64_t lane_off = (int64_t)h * D + dchunk * 8;
    if constexpr (MODE == 5) {                                                                   // timing probe
        if (slots[lane] + (int)st.q[0][0] + (int)st.q[G - 1][3] == 0x7fffffff) p.ws_ml[0] = 1.f;
        return;
    }

    auto load_tile = [&](int it, u32x4 (&kt)[U], u32x4 (&vt)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t off = (int64_t)slots[(it * U + u) * NT + tsub] * p.row_elems + lane_off;
            if constexpr (MODE == 2) {                    // ablation: no K/V traffic
                kt[u] = u32x4{(uint32_t)off, 0x3f803f80u, 0x3f803f80u, (uint32_t)it};
                vt[u] = kt[u];
            } else {
                kt[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p.k_pool + off));   // read once per step: see glds16p
                vt[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p.v_pool + off));
            }
        }
    };
    auto run_tile = [&](int it, const u32x4 (&kt)[U], const u32x4 (&vt)[U]) {
        if constexpr (MODE == 1) {                        // ablation: stream only, keep every load alive
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int w = 0; w < 4; ++w) st.q[0][w] ^= kt[u][w] ^ vt[u][w];
            return;
        }
        float bias[U];
#pragma unroll
        for (int u = 0; u < U; ++u) bias[u] = ((it * U + u) * NT + tsub) < ntok ? 0.f : kNegBig;
        compute_tile<D, G, U>(st, kt, vt, bias, p.scale_log2);
    };

    if constexpr (MODE == 6) { asm volatile("" : "+v"(st.q[0][0]), "+v"(st.q[G - 1][3])); stamp(2); }
    u32x4 kb[NB][U], vb[NB][U];
#pragma unroll
    for (int j = 0; j < NB - 1; ++j)
        if (j < n_it) load_tile(j, kb[j], vb[j]);
    if constexpr (MODE == 6) { asm volatile("" : "+v"(kb[0][0][0])); stamp(3); }
    for (int base = 0; base < (MODE == 3 ? 1 : n_it); base += NB) {
#pragma unroll
        for (int j = 0; j < NB; ++j) {                   // j is a compile-time constant: registers, not scratch
            const int it = base + j;
            if (it < n_it) {
                if (it + NB - 1 < n_it) load_tile(it + NB - 1, kb[(j + NB - 1) % NB], vb[(j + NB - 1) % NB]);
                run_tile(it, kb[j], vb[j]);
            }
        }
    }
    if constexpr (MODE == 1) st.o[0][0] = __builtin_bit_cast(float, st.q[0][0] ^ st.q[0][1] ^ st.q[0][2] ^ st.q[0][3]);
    if constexpr (MODE == 6) { asm volatile("" : "+v"(st.o[0][0])); stamp(4); }
    finish_wave<D, G>(p, st, qi, h, split, L, lane);
    if constexpr (MODE == 6) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp(5); }
}

// ------------------------------------------------------------------------------------------------
// Variant L: K/V rows by LDS-DMA (global_load_lds_dwordx4: per-lane source row, wave-linear 1 KiB LDS image)
// into a per-wave ring of R tiles.  R-1 tiles stay in flight per wave at no VGPR cost; each lane reads back
// exactly the 16 bytes it fetched (ds_read_b128 at lane*16: conflict-free), so no barrier is needed - only the
// issuing wave's own counted vmcnt.  The DMA and its waits are inline asm (hipcc would drain vmcnt(0) before
// every LDS read otherwise), so the kernel issues no other vector-memory operation inside the loop, and every
// load hipcc does count (page table, q) is forced to retire before the first DMA.
// ------------------------------------------------------------------------------------------------

template <int D, int G, int U, int R, int W, bool POW2>
__global__ __launch_bounds__(64, W) void decode_split_lds_kernel(const DecodeParams p) {
    constexpr int TILE = U * 2048;                       // bytes: U x (1 KiB of K rows + 1 KiB of V rows)
    __shared__ __attribute__((aligned(16))) uint32_t lds[(R * TILE) / 4 + kMaxChunk + 64];
    int32_t* slots = reinterpret_cast<int32_t*>(lds + (R * TILE) / 4);
    M355_WORK_ITEM_PROLOGUE()
    stage_slots<POW2>(p, slots, lane, row, t0, ntok, n_it * TPI);

    WaveState<D, G> st;
    init_state<D, G>(p, st, qi, h, dchunk);
    // A use of every q register here makes hipcc place its own vmcnt(0) for the q loads BEFORE the loop; left to
    // itself it waits for them at their first use inside the loop, which drains the DMA ring every iteration.
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int w = 0; w < 4; ++w) asm volatile("" : "+v"(st.q[g][w]));
    __syncthreads();

    const uint32_t ring = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds);
    const uint16_t* kbase = p.k_pool + (int64_t)h * D + dchunk * 8;
    const uint16_t* vbase = p.v_pool + (int64_t)h * D + dchunk * 8;

    auto issue = [&](int it) {
        const uint32_t dst = ring + (uint32_t)(it % R) * TILE;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t off = (int64_t)slots[(it * U + u) * NT + tsub] * p.row_elems;
            glds16p<1>(kbase + off, dst + u * 2048);
            glds16p<1>(vbase + off, dst + u * 2048 + 1024);
        }
    };
    auto consume = [&](int it) {
        const uint8_t* tile = reinterpret_cast<const uint8_t*>(lds) + (it % R) * TILE + lane * 16;
        u32x4 kt[U], vt[U];
        float bias[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            kt[u] = *reinterpret_cast<const u32x4*>(tile + u * 2048);
            vt[u] = *reinterpret_cast<const u32x4*>(tile + u * 2048 + 1024);
            bias[u] = ((it * U + u) * NT + tsub) < ntok ? 0.f : kNegBig;
        }
        compute_tile<D, G, U>(st, kt, vt, bias, p.scale_log2);
    };

    for (int j = 0; j < R - 1 && j < n_it; ++j) issue(j);
    int it = 0;
    for (; it + (R - 1) < n_it; ++it) {                  // steady state: R-1 younger tiles stay in flight
        issue(it + R - 1);
        wait_vmcnt<(R - 1) * 2 * U>();
        consume(it);
    }
    for (; it < n_it; ++it) {                            // drain: n_it-1-it younger tiles
        wait_tiles_younger<R - 2, 2 * U>(n_it - 1 - it);
        consume(it);
    }
    finish_wave<D, G>(p, st, qi, h, split, L, lane);
}

// One wave per (query row, q head): combine the chunk partials of sequences longer than one chunk.
template <int D>
__global__ __launch_bounds__(64) void decode_merge_kernel(const DecodeParams p) {
    const int qi = blockIdx.x / p.Hq;
    const int hq = blockIdx.x % p.Hq;
    int64_t L, row;
    query_row_len(p, qi, row, L);
    const int nsplit = (int)((L + p.chunk - 1) / p.chunk);
    if (nsplit <= 1) return;
    constexpr int E = D / kWave;                           // 1 or 2 elements per lane
    const int lane = threadIdx.x;
    const int64_t base = ((int64_t)qi * p.Hq + hq) * p.max_splits;
    float mx = kNegBig;
    for (int s = 0; s < nsplit; ++s) mx = fmaxf(mx, p.ws_ml[(base + s) * 2]);
    float den = 0.f, acc[E];
#pragma unroll
    for (int e = 0; e < E; ++e) acc[e] = 0.f;
    for (int s = 0; s < nsplit; ++s) {
        const float w = exp2_fast(p.ws_ml[(base + s) * 2] - mx);
        den += w * p.ws_ml[(base + s) * 2 + 1];
        const float* wo = p.ws_o + (base + s) * D + lane * E;
#pragma unroll
        for (int e = 0; e < E; ++e) acc[e] += w * wo[e];
    }
    const float inv = 1.f / den;
    uint16_t* op = p.out + (int64_t)qi * p.o_stride + (int64_t)hq * D + lane * E;
    if constexpr (E == 2) {
        *reinterpret_cast<uint32_t*>(op) = pack_bf16(acc[0] * inv, acc[1] * inv);
    } else {
        const __bf16 b = (__bf16)(acc[0] * inv);
        *op = __builtin_bit_cast(uint16_t, b);
    }
}

static int ilog2_exact(int32_t x) {
    if (x <= 0 || (x & (x - 1))) return -1;
    int s = 0;
    while ((1 << s) < x) ++s;
    return s;
}

// chunk (tokens per wave) heuristic: the largest power of two that still yields >= ~8 waves per CU,
// bounded to [128, kMaxChunk]; tiny problems go down to 64 tokens to get any parallelism at all.
static int32_t auto_chunk(int64_t nq, int32_t Hkv, int32_t max_len) {
    const int64_t want = 2048;
    for (int32_t c = kMaxChunk; c >= 128; c >>= 1) {
        const int64_t items = nq * Hkv * ((max_len + c - 1) / c);
        if (items >= want) return c;
    }
    return (nq * Hkv * ((max_len + 127) / 128) >= 512) ? 128 : 64;
}


// Variant selection.  M355_DECODE_VARIANT (read once) is a tuning hook: 0 = default.
static int decode_variant() {
    static int v = [] { const char* e = getenv("M355_DECODE_VARIANT"); return e ? atoi(e) : 0; }();
    return v;
}

template <int D, int G, bool POW2>
static void launch_decode(const DecodeParams& p, hipStream_t st) {
    const uint32_t blocks = (uint32_t)((int64_t)p.NQ * p.Hkv * p.max_splits);
    const dim3 g(blocks), b(64);
    switch (decode_variant()) {
        //                                                       D  G  U NB/R W
        case 1: hipLaunchKernelGGL((decode_split_kernel<D, G, 2, 2, 3, POW2>), g, b, 0, st, p); break;
        case 2: hipLaunchKernelGGL((decode_split_kernel<D, G, 2, 3, 3, POW2>), g, b, 0, st, p); break;
        case 3: hipLaunchKernelGGL((decode_split_kernel<D, G, 2, 4, 2, POW2>), g, b, 0, st, p); break;
        case 4: hipLaunchKernelGGL((decode_split_lds_kernel<D, G, 2, 4, 2, POW2>), g, b, 0, st, p); break;
        case 5: hipLaunchKernelGGL((decode_split_lds_kernel<D, G, 2, 6, 2, POW2>), g, b, 0, st, p); break;
        case 6: hipLaunchKernelGGL((decode_split_lds_kernel<D, G, 1, 8, 2, POW2>), g, b, 0, st, p); break;
        case 7: hipLaunchKernelGGL((decode_split_lds_kernel<D, G, 4, 3, 2, POW2>), g, b, 0, st, p); break;
        case 8: hipLaunchKernelGGL((decode_split_kernel<D, G, 1, 6, 3, POW2>), g, b, 0, st, p); break;
        case 9: hipLaunchKernelGGL((decode_split_kernel<D, G, 4, 3, 2, POW2>), g, b, 0, st, p); break;
        case 11: hipLaunchKernelGGL((decode_split_kernel<D, G, 2, 3, 3, POW2, 1>), g, b, 0, st, p); break;   // loads only
        case 12: hipLaunchKernelGGL((decode_split_kernel<D, G, 2, 3, 3, POW2, 2>), g, b, 0, st, p); break;   // compute only
        case 14: hipLaunchKernelGGL((decode_split_kernel<D, G, 2, 3, 3, POW2, 3>), g, b, 0, st, p); break;   // one pass of NB tiles only
        case 15: hipLaunchKernelGGL((decode_split_kernel<D, G, 2, 3, 3, POW2, 4>), g, b, 0, st, p); break;   // exit after lens/row
        case 16: hipLaunchKernelGGL((decode_split_kernel<D, G, 2, 3, 3, POW2, 5>), g, b, 0, st, p); break;   // exit after slots + q
        case 17: hipLaunchKernelGGL((decode_split_kernel<D, G, 2, 3, 3, POW2, 6>), g, b, 0, st, p); break;   // stamped
        case 13: hipLaunchKernelGGL((decode_split_kernel<D, G, 4, 3, 2, POW2, 1>), g, b, 0, st, p); break;   // loads only, deeper
        default: hipLaunchKernelGGL((decode_split_kernel<D, G, (G >= 8 ? 2 : 4), 2, 2, POW2>), g, b, 0, st, p); break;
    }
    if (p.max_splits > 1)
        hipLaunchKernelGGL((decode_merge_kernel<D>), dim3((uint32_t)(p.NQ * p.Hq)), dim3(64), 0, st, p);
}

// Shared by m355_decode_attn (cu_q == null: one query per request) and by the row-parallel extend path of
// m355_prefill_attn (cu_q/prefix_lens given: NQ packed query rows, each treated as its own decode query).
int paged_rows_attn(const void* q, int64_t q_stride, const void* k_pool, const void* v_pool, int64_t pool_row_stride,
                    const int32_t* page_table,
                    int64_t pt_stride, const int64_t* row_ids, const int64_t* kv_lens, const int32_t* cu_q,
                    const int64_t* prefix_lens, int32_t B, void* out, int64_t o_stride, int64_t NQ, int32_t Hq,
                    int32_t Hkv, int32_t D, int32_t ps, float scale, int32_t max_len_hint, int32_t chunk,
                    void* workspace, int64_t workspace_bytes, int32_t dtype, hipStream_t st) {
    if (NQ == 0) return M355_OK;
    if (!q || !k_pool || !v_pool || !page_table || !row_ids || !out) return M355_ERR_INVALID_ARG;
    if (cu_q ? (!prefix_lens || B <= 0) : !kv_lens) return M355_ERR_INVALID_ARG;
    if (NQ < 0 || Hq <= 0 || Hkv <= 0 || ps <= 0 || max_len_hint <= 0 || Hq % Hkv) return M355_ERR_INVALID_ARG;
    if (dtype != M355_DTYPE_BF16) return M355_ERR_UNSUPPORTED;
    if (D != 64 && D != 128) return M355_ERR_UNSUPPORTED;
    const int G = Hq / Hkv;
    if (G != 1 && G != 2 && G != 4 && G != 8) return M355_ERR_UNSUPPORTED;
    if (((uintptr_t)q | (uintptr_t)k_pool | (uintptr_t)v_pool | (uintptr_t)out) & 15) return M355_ERR_INVALID_ARG;
    if ((q_stride % 8) || (o_stride % 8) || q_stride < (int64_t)Hq * D || o_stride < (int64_t)Hq * D)
        return M355_ERR_INVALID_ARG;
    if (chunk <= 0) chunk = auto_chunk(NQ, Hkv, max_len_hint);
    if (chunk > kMaxChunk || chunk % 64) return M355_ERR_INVALID_ARG;
    const int32_t max_splits = (max_len_hint + chunk - 1) / chunk;
    if ((int64_t)NQ * Hkv * max_splits > 0x7fffffffLL) return M355_ERR_INVALID_ARG;

    DecodeParams p;
    p.q = (const uint16_t*)q; p.k_pool = (const uint16_t*)k_pool; p.v_pool = (const uint16_t*)v_pool;
    p.page_table = page_table; p.row_ids = row_ids; p.kv_lens = kv_lens; p.out = (uint16_t*)out;
    p.cu_q = cu_q; p.prefix_lens = prefix_lens; p.B = B;
    p.q_stride = q_stride; p.o_stride = o_stride; p.pt_stride = pt_stride; p.row_elems = pool_row_stride ? pool_row_stride : (int64_t)Hkv * D;
    p.NQ = (int32_t)NQ; p.Hq = Hq; p.Hkv = Hkv; p.ps = ps; p.ps_shift = ilog2_exact(ps);
    p.chunk = chunk; p.max_splits = max_splits; p.scale_log2 = scale * kLog2e;
    p.ws_o = nullptr; p.ws_ml = nullptr;
    if (max_splits > 1) {
        const int64_t n = NQ * Hq * max_splits;
        const int64_t need = n * (D + 2) * (int64_t)sizeof(float);
        if (!workspace || workspace_bytes < need || ((uintptr_t)workspace & 15)) return M355_ERR_WORKSPACE;
        p.ws_o = (float*)workspace;
        p.ws_ml = p.ws_o + n * D;
    }
    const bool pow2 = p.ps_shift >= 0;
#define M355_CASE(DD, GG) if (D == DD && G == GG) { if (pow2) launch_decode<DD, GG, true>(p, st); else launch_decode<DD, GG, false>(p, st); return check_launch(); }
    M355_CASE(64, 1) M355_CASE(64, 2) M355_CASE(64, 4) M355_CASE(64, 8)
    M355_CASE(128, 1) M355_CASE(128, 2) M355_CASE(128, 4) M355_CASE(128, 8)
#undef M355_CASE
    return M355_ERR_UNSUPPORTED;
}

int32_t decode_pick_chunk(int64_t nq, int32_t Hkv, int32_t max_len, int32_t chunk) {
    return chunk > 0 ? chunk : auto_chunk(nq, Hkv, max_len);
}

}  // namespace m355

extern "C" int m355dbg_set_stamp_buffer(void* dev_ptr) {      // dev-only (not part of include/mi355attn.h)
    return hipMemcpyToSymbol(HIP_SYMBOL(m355::g_stamp_buf), &dev_ptr, sizeof(dev_ptr)) == hipSuccess ? 0 : -1;
}

extern "C" int64_t m355dbg_decode_chunked_workspace_bytes(int32_t B, int32_t Hq, int32_t D, int32_t max_seq_len_hint,
                                               int32_t chunk_tokens) {
    if (B <= 0 || Hq <= 0 || D <= 0 || max_seq_len_hint <= 0) return 0;
    // Hkv is not known here; size for the smallest chunk the auto rule can pick unless one is given
    const int32_t chunk = chunk_tokens > 0 ? chunk_tokens : 64;
    const int64_t splits = (max_seq_len_hint + chunk - 1) / chunk;
    if (splits <= 1) return 0;
    return (int64_t)B * Hq * splits * (D + 2) * (int64_t)sizeof(float);
}

extern "C" int m355dbg_decode_attn_chunked(const void* q, int64_t q_row_stride, const void* k_pool, const void* v_pool,
                                int64_t pool_row_stride, const int32_t* page_table, int64_t pt_row_stride, const int64_t* page_table_ids,
                                const int64_t* seq_lens, void* out, int64_t o_row_stride, int32_t B, int32_t Hq,
                                int32_t Hkv, int32_t D, int32_t page_size, float scale, int32_t max_seq_len_hint,
                                int32_t chunk_tokens, void* workspace, int64_t workspace_bytes, int32_t dtype,
                                void* stream) {
    if (pool_row_stride && (pool_row_stride < (int64_t)Hkv * D || pool_row_stride % 8)) return M355_ERR_INVALID_ARG;
    return m355::paged_rows_attn(q, q_row_stride, k_pool, v_pool, pool_row_stride, page_table, pt_row_stride, page_table_ids, seq_lens,
                                 nullptr, nullptr, B, out, o_row_stride, B, Hq, Hkv, D, page_size, scale,
                                 max_seq_len_hint, chunk_tokens, workspace, workspace_bytes, dtype, (hipStream_t)stream);
}
#endif  // M355_DEV
