// Ragged prefill ("extend") attention over cached prefixes (reference torch_native_backend.py:25-119, 195-235).
//
// MFMA-bound for long extends: 4*Hq*D*sum(E*P + E*(E+1)/2) FLOPs against the bf16 dense peak; HBM-bound
// (4*Hkv*D*sum(L) + 4*T*Hq*D bytes) for short ones.
//
// One wave = one tile of 32 query ROWS of one (request, kv head).  A query row is a (token, q-head-of-the-group)
// pair, token-major, so the G = Hq/Hkv heads that share a kv head sit in the same tile and share every K/V load
// (for G = 4 a tile is 8 tokens x 4 heads).  Per 32-token KV tile:
//   S^T[token, row]  = K[32 x D] . Q^T[D x 32]     v_mfma_f32_32x32x16_bf16, K rows straight from HBM into the A
//                      operand (a lane's 8-element fragment is 16 contiguous bytes of one K row), Q^T held in
//                      registers.  "Swapped" product: a lane then owns ONE query row (column lane&31) and 16 of the
//                      32 tokens in its accumulator registers, so the online softmax is lane-local except for one
//                      v_permlane32_swap with the lane holding the other 16 tokens.
//   O^T[d, row]     += V^T[D x 32] . P^T[32 x 32]  the bf16-converted S^T accumulator IS the B operand (no lane
//                      movement); V rows are staged row-major in LDS by coalesced 16-byte loads and read back
//                      transposed with ds_read_b64_tr_b16 (row stride D*2 + 64 bytes: conflict-free for both the
//                      128-bit writes and the transposed reads).
// The causal mask is offset by the prefix length: row (token j of the extend) sees cached tokens 0 .. prefix + j.
// K/V of the new tokens are read back from the pool (the KV write precedes this kernel on the stream).
#include <stdlib.h>

#include "common.h"
#include "paged_rows.h"

namespace m355 {

struct PrefillParams {
    const uint16_t* q;
    const uint16_t* k_pool;
    const uint16_t* v_pool;
    const int32_t* page_table;
    const int64_t* page_table_ids;
    const int64_t* seq_lens;
    const int64_t* prefix_lens;
    const int32_t* cu_q;
    uint16_t* out;
    int64_t q_stride, o_stride, pt_stride, row_elems;
    int32_t B, Hq, Hkv, ps, ps_shift, tiles_max;
    float scale_log2;
};

template <int D>
struct PrefillCfg {
    static constexpr int KS = D / 16;                 // k-steps of the QK^T product (16 head-dim elements each)
    static constexpr int DT = D / 32;                 // 32-row tiles of O^T
    static constexpr int LPH = D / 8;                 // lanes per V row (16 bytes each)
    static constexpr int NT = kWave / LPH;            // V rows per wave-wide load
    static constexpr int VL = 32 / NT;                // wave-wide loads per 32-token V tile
    static constexpr int VROW = D * 2 + 64;           // LDS row stride in bytes
    static constexpr int VTILE = 32 * VROW;           // LDS bytes per wave
};

template <int D, int G, bool POW2>
__global__ __launch_bounds__(256) void prefill_mfma_kernel(const PrefillParams p) {
    using C = PrefillCfg<D>;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int l = threadIdx.x & 63;
    const int c = l & 31, hh = l >> 5;
    uint8_t* vl = smem + wid * C::VTILE;

    // ---- which tile
    int item = blockIdx.x * 4 + wid;
    const int t = item % p.tiles_max;  item /= p.tiles_max;
    const int h = item % p.Hkv;
    const int b = item / p.Hkv;
    if (b >= p.B) return;
    const int P = (int)p.prefix_lens[b];
    const int E = (int)p.seq_lens[b] - P;
    const int rows_b = E * G;
    if (32 * t >= rows_b) return;

    const int m = 32 * t + c;                             // this lane's query row
    const bool valid_col = m < rows_b;
    const int j = m / G, g = m % G;
    const int limit = valid_col ? P + j + 1 : 0;          // tokens this row attends
    const int j_last = min(E - 1, (32 * t + 31) / G);
    const int kv_end = P + j_last + 1;                    // tokens any row of the tile attends
    const int kv_min = P + (32 * t) / G + 1;              // tokens EVERY valid row attends
    const int n_tiles = (kv_end + 31) >> 5;
    const int32_t* pt_row = p.page_table + p.page_table_ids[b] * p.pt_stride;
    const int64_t qrow0 = p.cu_q[b];

    // ---- Q^T fragments: lane (row c, half hh), k-step s <-> head-dim elements 8*(hh*KS + s) .. +8.
    // The K fragments use the same element order; the dot product does not care which order it is summed in.
    bf16x8 qf[C::KS];
    {
        const uint16_t* qp = p.q + (qrow0 + j) * p.q_stride + (int64_t)(h * G + g) * D + hh * (D / 2);
#pragma unroll
        for (int s = 0; s < C::KS; ++s) {
            u32x4 v = u32x4{0u, 0u, 0u, 0u};
            if (valid_col) v = *reinterpret_cast<const u32x4*>(qp + s * 8);
            qf[s] = __builtin_bit_cast(bf16x8, v);
        }
    }

    auto slot_of = [&](int tile) -> int32_t {            // slot of token 32*tile + c (clamped into the request)
        int tok = tile * 32 + c;
        tok = tok < kv_end ? tok : kv_end - 1;
        const int pg = POW2 ? (tok >> p.ps_shift) : (tok / p.ps);
        return pt_row[pg] * p.ps + (tok - (POW2 ? (pg << p.ps_shift) : pg * p.ps));
    };
    bf16x8 kf[C::KS];
    u32x4 vreg[C::VL];
    auto load_kv = [&](int32_t slot) {                   // K fragments + V rows of the tile whose slots the lanes hold
        const uint16_t* kp = p.k_pool + (int64_t)slot * p.row_elems + h * D + hh * (D / 2);
#pragma unroll
        for (int s = 0; s < C::KS; ++s) kf[s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(kp + s * 8));
#pragma unroll
        for (int u = 0; u < C::VL; ++u) {
            const int tv = u * C::NT + l / C::LPH;        // V row (token of the tile) this lane helps to fetch
            const int32_t sv = __shfl(slot, tv);
            vreg[u] = *reinterpret_cast<const u32x4*>(p.v_pool + (int64_t)sv * p.row_elems + h * D + (l % C::LPH) * 8);
        }
    };
    auto stage_v = [&]() {
#pragma unroll
        for (int u = 0; u < C::VL; ++u) {
            const int tv = u * C::NT + l / C::LPH;
            *reinterpret_cast<u32x4*>(vl + tv * C::VROW + (l % C::LPH) * 16) = vreg[u];
        }
    };

    f32x16 o[C::DT];
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[dt][i] = 0.f;
    float m_run = kNegBig, l_run = 0.f;

    // transposed-read addresses (constant per lane): 16-lane group gidx reads a 4-token x 16-column block
    const int gidx = l >> 4, li = l & 15;
    const int tr_base = (4 * hh + (li >> 2)) * C::VROW + (16 * (gidx & 1) + 4 * (li & 3)) * 2;

    int32_t slot_cur = slot_of(0);
    int32_t slot_nxt = n_tiles > 1 ? slot_of(1) : slot_cur;
    load_kv(slot_cur);
    stage_v();

    for (int n = 0; n < n_tiles; ++n) {
        // ---- S^T = K . Q^T
        f32x16 s;
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = 0.f;
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[ks], s, 0, 0, 0);

        // ---- next tile's K/V go out now (K registers are free again), the slots of the one after that too
        const bool more = n + 1 < n_tiles;
        if (more) {
            load_kv(slot_nxt);
            slot_nxt = (n + 2 < n_tiles) ? slot_of(n + 2) : slot_nxt;
        }

        // ---- online softmax; this lane: row c, tokens 32n + (i&3) + 8*(i>>2) + 4*hh
        const int base = n * 32;
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] *= p.scale_log2;
        if (base + 32 > kv_min) {                         // wave-uniform: only the diagonal tiles pay for the mask
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int tok = base + (i & 3) + 8 * (i >> 2) + 4 * hh;
                s[i] = tok < limit ? s[i] : kNegBig;
            }
        }
        float mx = s[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) mx = fmaxf(mx, s[i]);
        float ma, mb;
        lane_pair<32>(mx, ma, mb);
        const float m_new = fmaxf(m_run, fmaxf(ma, mb));
        const float alpha = exp2_fast(m_run - m_new);
        m_run = m_new;
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) { s[i] = exp2_fast(s[i] - m_new); sum += s[i]; }
        l_run = fmaf(l_run, alpha, sum);
        if (__any(alpha != 1.f)) {
#pragma unroll
            for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
                for (int i = 0; i < 16; ++i) o[dt][i] *= alpha;
        }
        bf16x8 pf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            u32x4 w;
#pragma unroll
            for (int x = 0; x < 4; ++x) w[x] = pack_bf16(s[8 * ks + 2 * x], s[8 * ks + 2 * x + 1]);
            pf[ks] = __builtin_bit_cast(bf16x8, w);
        }

        // ---- O^T += V^T . P^T with V^T read transposed from LDS
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const uint8_t* a0 = vl + tr_base + (16 * ks) * C::VROW + dt * 64;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (s16x4 __attribute__((address_space(3)))*)(uintptr_t)(uint32_t)(uintptr_t)a0);
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (s16x4 __attribute__((address_space(3)))*)(uintptr_t)(uint32_t)(uintptr_t)(a0 + 8 * C::VROW));
                const s16x8 av = s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), pf[ks], o[dt], 0, 0, 0);
            }

        // ---- the next V tile replaces this one in LDS (same wave: LDS operations execute in order)
        if (more) stage_v();
    }

    // ---- normalise and store: lane (row c, half hh) holds d = 32*dt + 8*(i>>2) + 4*hh + (i&3)
    float la, lb;
    lane_pair<32>(l_run, la, lb);
    const float inv = 1.f / (la + lb);
    if (valid_col) {
        uint16_t* op = p.out + (qrow0 + j) * p.o_stride + (int64_t)(h * G + g) * D + 4 * hh;
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                uint2 w;
                w.x = pack_bf16(o[dt][4 * q4 + 0] * inv, o[dt][4 * q4 + 1] * inv);
                w.y = pack_bf16(o[dt][4 * q4 + 2] * inv, o[dt][4 * q4 + 3] * inv);
                *reinterpret_cast<uint2*>(op + 32 * dt + 8 * q4) = w;
            }
    }
}

static int ilog2_exact_p(int32_t x) {
    if (x <= 0 || (x & (x - 1))) return -1;
    int s = 0;
    while ((1 << s) < x) ++s;
    return s;
}

static int prefill_mode() {      // tuning / cross-check hook: 1 = force the row-parallel (VALU) path
    static int v = [] { const char* e = getenv("M355_PREFILL_MODE"); return e ? atoi(e) : 0; }();
    return v;
}

template <int D, int G>
static void launch_prefill(const PrefillParams& p, hipStream_t st) {
    const int64_t items = (int64_t)p.B * p.Hkv * p.tiles_max;
    const dim3 grid((unsigned)((items + 3) / 4)), block(256);
    const size_t lds = 4 * PrefillCfg<D>::VTILE;
    if (p.ps_shift >= 0) hipLaunchKernelGGL((prefill_mfma_kernel<D, G, true>), grid, block, lds, st, p);
    else hipLaunchKernelGGL((prefill_mfma_kernel<D, G, false>), grid, block, lds, st, p);
}

}  // namespace m355

using namespace m355;

extern "C" int64_t m355_prefill_workspace_bytes(int32_t T, int32_t Hq, int32_t D, int32_t max_seq_len_hint) {
    if (T <= 0 || Hq <= 0 || D <= 0 || max_seq_len_hint <= 0) return 0;
    if (prefill_mode() != 1) return 0;                   // the MFMA path keeps everything on chip
    const int64_t splits = (max_seq_len_hint + kRowsMaxChunk - 1) / kRowsMaxChunk;
    if (splits <= 1) return 0;
    return (int64_t)T * Hq * splits * (D + 2) * (int64_t)sizeof(float);
}

extern "C" int m355_prefill_attn(const void* q, int64_t q_row_stride, const void* k_pool, const void* v_pool,
                                 const int32_t* page_table, int64_t pt_row_stride, const int64_t* page_table_ids,
                                 const int64_t* seq_lens, const int64_t* prefix_lens, const int32_t* cu_q, void* out,
                                 int64_t o_row_stride, int32_t B, int32_t T, int32_t max_extend_len,
                                 int32_t max_seq_len_hint, int32_t Hq, int32_t Hkv, int32_t D, int32_t page_size,
                                 float scale, void* workspace, int64_t workspace_bytes, int32_t dtype, void* stream) {
    if (T == 0 || B == 0) return M355_OK;
    if (!q || !k_pool || !v_pool || !page_table || !out || !seq_lens || !prefix_lens || !cu_q || !page_table_ids || B < 0 ||
        T < 0 || max_extend_len <= 0 || Hq <= 0 || Hkv <= 0 || Hq % Hkv || page_size <= 0)
        return M355_ERR_INVALID_ARG;
    if (prefill_mode() == 1)        // row-parallel path: every packed query row handled like a decode query
        return paged_rows_attn(q, q_row_stride, k_pool, v_pool, page_table, pt_row_stride, page_table_ids, nullptr, cu_q,
                               prefix_lens, B, out, o_row_stride, T, Hq, Hkv, D, page_size, scale, max_seq_len_hint,
                               kRowsMaxChunk, workspace, workspace_bytes, dtype, (hipStream_t)stream);
    if (dtype != M355_DTYPE_BF16) return M355_ERR_UNSUPPORTED;
    if (D != 64 && D != 128) return M355_ERR_UNSUPPORTED;
    const int G = Hq / Hkv;
    if (G != 1 && G != 2 && G != 4 && G != 8) return M355_ERR_UNSUPPORTED;
    if (((uintptr_t)q | (uintptr_t)k_pool | (uintptr_t)v_pool | (uintptr_t)out) & 15) return M355_ERR_INVALID_ARG;
    if ((q_row_stride % 8) || (o_row_stride % 8) || q_row_stride < (int64_t)Hq * D || o_row_stride < (int64_t)Hq * D)
        return M355_ERR_INVALID_ARG;
    PrefillParams p;
    p.q = (const uint16_t*)q; p.k_pool = (const uint16_t*)k_pool; p.v_pool = (const uint16_t*)v_pool;
    p.page_table = page_table; p.page_table_ids = page_table_ids; p.seq_lens = seq_lens; p.prefix_lens = prefix_lens;
    p.cu_q = cu_q; p.out = (uint16_t*)out;
    p.q_stride = q_row_stride; p.o_stride = o_row_stride; p.pt_stride = pt_row_stride; p.row_elems = (int64_t)Hkv * D;
    p.B = B; p.Hq = Hq; p.Hkv = Hkv; p.ps = page_size; p.ps_shift = ilog2_exact_p(page_size);
    p.tiles_max = (max_extend_len * G + 31) / 32;
    p.scale_log2 = scale * kLog2e;
    if ((int64_t)B * Hkv * p.tiles_max > 0x7fffffffLL) return M355_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
#define M355_CASE(DD, GG) if (D == DD && G == GG) { launch_prefill<DD, GG>(p, st); return check_launch(); }
    M355_CASE(64, 1) M355_CASE(64, 2) M355_CASE(64, 4) M355_CASE(64, 8)
    M355_CASE(128, 1) M355_CASE(128, 2) M355_CASE(128, 4) M355_CASE(128, 8)
#undef M355_CASE
    return M355_ERR_UNSUPPORTED;
}
