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

// ------------------------------------------------------------------------------------------------
// Workgroup-shared variant (default): the 4 waves of a workgroup own 4*RT consecutive row tiles of one
// (request, kv head) and walk the request's KV together.  Every 32-token K/V tile is brought into LDS ONCE per
// workgroup with global_load_lds_dwordx4 (whole 2*D-byte rows, XOR-swizzled through the choice of the chunk each
// lane fetches, double buffered) and consumed by all waves: K fragments are ds_read_b128 rows, V^T fragments
// ds_read_b64_tr_b16 transposed reads.  L2->CU traffic drops by 4*RT against one-wave-per-tile streaming, which was
// what bounded the first kernel (3.3 GB per layer on the 8B shared-prefix workload).
// ------------------------------------------------------------------------------------------------
template <int D>
struct WgCfg {
    static constexpr int C = D / 8;                   // 16-byte chunks per row
    static constexpr int ROWB = 2 * D;
    static constexpr int RPI = 64 / C;                // rows per DMA instruction
    static constexpr int NI = 32 / RPI;               // DMA instructions per K (or V) tile
    static constexpr int IPW = 2 * NI / 4;            // DMA instructions per wave per KV tile (K and V)
    static constexpr int TILEB = 32 * ROWB;
    static constexpr int BUFB = 2 * TILEB;            // K tile then V tile
    static constexpr int KS = D / 16, DT = D / 32;
    static constexpr int SPAN = 16 / C;               // rows per 256-byte bank span
    // chunk ch of row `row` sits at position ch ^ s(row): conflict-free 16-row ds_read_b128 column reads of K,
    // conflict-free 4-row x 64-byte transposed reads of V
    __device__ static constexpr int sK(int row) { return (row / SPAN) % C; }
    __device__ static constexpr int sV(int row) { return ((row / SPAN) % (C / 4)) * 4; }
};

constexpr float kLazy = 8.f;                          // log2 units
constexpr int kSlotWin = 1024;                        // tokens whose pool slots are staged in LDS at a time

template <int D, int G, int RT, int NBUF, bool POW2>
__global__ __launch_bounds__(256, RT == 1 ? 2 : 1) void prefill_wg_kernel(const PrefillParams p) {
    using W = WgCfg<D>;
    extern __shared__ __attribute__((aligned(1024))) uint8_t smem[];     // NBUF K tiles, NBUF V tiles, the slot window
    int32_t* const slotw = reinterpret_cast<int32_t*>(smem + NBUF * W::BUFB);
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int l = threadIdx.x & 63;
    const int c = l & 31, hh = l >> 5;
    constexpr int ROWS_WG = 128 * RT;

    int item = blockIdx.x;
    const int tb = p.tiles_max - 1 - item % p.tiles_max;  item /= p.tiles_max;   // row block; the longest (latest) first
    const int h = item % p.Hkv;
    const int b = item / p.Hkv;
    const int P = (int)p.prefix_lens[b];
    const int E = (int)p.seq_lens[b] - P;
    const int rows_b = E * G;
    if (ROWS_WG * tb >= rows_b) return;                            // whole workgroup: no barrier has been reached yet
    const int32_t* pt_row = p.page_table + p.page_table_ids[b] * p.pt_stride;
    const int64_t qrow0 = p.cu_q[b];
    const int j_last_wg = min(E - 1, (ROWS_WG * tb + ROWS_WG - 1) / G);
    const int kv_end_wg = P + j_last_wg + 1;                       // tokens any row of the workgroup attends
    const int n_tiles = (kv_end_wg + 31) >> 5;

    // ---- per row tile: the lane's row, its limit, Q^T fragments
    int limit[RT], kv_end[RT], kv_min[RT], jrow[RT], grow[RT];
    bool valid[RT];
    bf16x8 qf[RT][W::KS];
    f32x16 o[RT][W::DT];
    float m_run[RT], l_run[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        const int t = (tb * 4 + wid) * RT + r;                     // row tile index inside the request
        const int m = 32 * t + c;
        valid[r] = m < rows_b;
        jrow[r] = m / G; grow[r] = m % G;
        limit[r] = valid[r] ? P + jrow[r] + 1 : 0;
        kv_end[r] = 32 * t < rows_b ? P + min(E - 1, (32 * t + 31) / G) + 1 : 0;
        kv_min[r] = P + (32 * t) / G + 1;
        const uint16_t* qp = p.q + (qrow0 + jrow[r]) * p.q_stride + (int64_t)(h * G + grow[r]) * D + hh * 8;
#pragma unroll
        for (int s = 0; s < W::KS; ++s) {
            u32x4 v = u32x4{0u, 0u, 0u, 0u};
            if (valid[r]) v = *reinterpret_cast<const u32x4*>(qp + s * 16);      // d = 16 s + 8 hh + j: the operand's own order
            qf[r][s] = __builtin_bit_cast(bf16x8, v);
        }
#pragma unroll
        for (int dt = 0; dt < W::DT; ++dt)
#pragma unroll
            for (int i = 0; i < 16; ++i) o[r][dt][i] = 0.f;
        m_run[r] = kNegBig; l_run[r] = 0.f;
    }

    // ---- DMA duty of this wave: instruction ids wid, wid + 4, ... of the tile's 2*NI (K first, then V)
    const int drow = l / W::C, dch = l % W::C;
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)smem);
    const uint8_t* kbase = reinterpret_cast<const uint8_t*>(p.k_pool) + (int64_t)h * W::ROWB;
    const uint8_t* vbase = reinterpret_cast<const uint8_t*>(p.v_pool) + (int64_t)h * W::ROWB;
    const int64_t row_bytes = p.row_elems * 2;
    // slots of tokens [win*kSlotWin, +kSlotWin) -> LDS (ordinary loads: the compiler's wait drains the DMA queue too,
    // once per 32 KV tiles)
    auto fill_slots = [&](int win) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kSlotWin / 256; ++k) {
            int tok = win * kSlotWin + (int)threadIdx.x + 256 * k;
            tok = tok < kv_end_wg ? tok : kv_end_wg - 1;
            const int pg = POW2 ? (tok >> p.ps_shift) : (tok / p.ps);
            slotw[threadIdx.x + 256 * k] = pt_row[pg] * p.ps + (tok - (POW2 ? (pg << p.ps_shift) : pg * p.ps));
        }
        __syncthreads();
    };
    auto issue = [&](int tile) {                                   // this wave's share of KV tile `tile`
        if ((tile & (kSlotWin / 32 - 1)) == 0) fill_slots(tile / (kSlotWin / 32));   // workgroup-uniform
        const int buf = tile % NBUF;
#pragma unroll
        for (int u = 0; u < W::IPW; ++u) {
            const int id = wid + 4 * u;                            // wave-uniform
            const bool is_v = id >= W::NI;
            const int i = id % W::NI;
            const int row = i * W::RPI + drow;
            const int ch = dch ^ (is_v ? W::sV(row) : W::sK(row));
            const int32_t slot = slotw[((tile * 32) & (kSlotWin - 1)) + row];
            const uint8_t* src = (is_v ? vbase : kbase) + (int64_t)slot * row_bytes + 16 * ch;
            glds16(src, lds0 + (is_v ? NBUF * W::TILEB : 0) + buf * W::TILEB + 1024 * i);
        }
    };

    // ---- LDS read addresses (per lane constants)
    // K fragment (token row c, k-step s): logical chunk 2 s + hh of row c.  V^T block: 16-lane group gidx reads 4 token
    // rows x 16 d columns; element order matches the P^T operand.  All per-lane parts are computed once; buffer, k-step
    // row block and "+8 rows" are immediates of the ds_read instructions (K tiles sit below 32 KiB, V tiles below 64 KiB).
    uint32_t kaddr[W::KS], vaddr[W::DT];
    {
        const int ksw = W::sK(c);
#pragma unroll
        for (int s = 0; s < W::KS; ++s) kaddr[s] = lds0 + c * W::ROWB + 16 * ((2 * s + hh) ^ ksw);
        const int gidx = l >> 4, li = l & 15;
        const int vrow = 4 * hh + (li >> 2);
        const int vcol_chunk = 2 * (gidx & 1) + ((li & 3) >> 1);   // chunk inside the 64-byte d-tile column block
#pragma unroll
        for (int dt = 0; dt < W::DT; ++dt)
            vaddr[dt] = lds0 + NBUF * W::TILEB + vrow * W::ROWB + 16 * ((4 * dt + vcol_chunk) ^ W::sV(vrow)) + 8 * (li & 1);
    }
    static_assert(2 * NBUF * W::TILEB <= 65536, "ds_read immediates");

    wait_vmcnt<0>();                                               // q fragments are in: the DMA count starts clean
    static_assert(NBUF == 4, "two pairs of tile buffers");
    issue(0);
    if (n_tiles > 1) issue(1);

    // KV tiles are consumed in PAIRS so that one wave has two independent chains to interleave: the QK^T MFMAs of
    // the second tile run under the softmax (VALU) of the first, the PV MFMAs of the first under the softmax of the
    // second.  Per pair: wait for this wave's DMA, barrier (both tiles landed; everyone left the previous pair),
    // send the next pair into the buffers the previous pair vacated, compute.  (Keeping a K fragment set alive
    // across the barrier instead - a deeper software pipeline - cost the second wave per SIMD and lost.)
    auto read_k = [&](bf16x8 (&kf)[W::KS], const int buf) {
#pragma unroll
        for (int s = 0; s < W::KS; ++s)
            kf[s] = *reinterpret_cast<const bf16x8 __attribute__((address_space(3)))*>((uintptr_t)(kaddr[s] + buf * W::TILEB));
    };
    auto qk = [&](f32x16& sc, const bf16x8 (&kf)[W::KS], const int r) {
#pragma unroll
        for (int i = 0; i < 16; ++i) sc[i] = 0.f;
#pragma unroll
        for (int ks = 0; ks < W::KS; ++ks) sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[r][ks], sc, 0, 0, 0);
    };
    auto softmax = [&](f32x16& s, bf16x8 (&pf)[2], const int r, const int base) {
        if (base + 32 > kv_min[r]) {                               // wave-uniform: only diagonal tiles pay for the mask
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int tok = base + (i & 3) + 8 * (i >> 2) + 4 * hh;
                s[i] = tok < limit[r] ? s[i] : kNegBig;
            }
        }
        float mx = s[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) mx = fmaxf(mx, s[i]);
        float ma, mb;
        lane_pair<32>(mx, ma, mb);
        // lazy running max: move it only when the tile's max beats it by more than 2^kLazy (the rescale of the
        // O^T accumulators - AGPR round trips - is then rare); p <= 2^kLazy stays well inside bf16 / fp32 range
        const float m_tile = fmaxf(ma, mb) * p.scale_log2;         // scores stay raw; scale > 0 commutes with max
        const float m_new = m_tile > m_run[r] + kLazy ? m_tile : m_run[r];
        const float alpha = exp2_fast(m_run[r] - m_new);
        m_run[r] = m_new;
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) { s[i] = exp2_fast(fmaf(s[i], p.scale_log2, -m_new)); sum += s[i]; }
        l_run[r] = fmaf(l_run[r], alpha, sum);
        if (__any(alpha != 1.f)) {
#pragma unroll
            for (int dt = 0; dt < W::DT; ++dt)
#pragma unroll
                for (int i = 0; i < 16; ++i) o[r][dt][i] *= alpha;
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            u32x4 w;
#pragma unroll
            for (int x = 0; x < 4; ++x) w[x] = pack_bf16(s[8 * ks + 2 * x], s[8 * ks + 2 * x + 1]);
            pf[ks] = __builtin_bit_cast(bf16x8, w);
        }
    };
    // O^T += V^T P^T for every active row tile of the wave: each V^T fragment is read from LDS once and feeds RT MFMAs
    auto pv = [&](const bf16x8 (&pf)[RT][2], const bool (&act)[RT], const int buf) {
#pragma unroll
        for (int dt = 0; dt < W::DT; ++dt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                // token rows 16 ks + 4 hh + q and + 8 (same swizzle: it only depends on the row modulo 8)
                const uint32_t a0 = vaddr[dt] + buf * W::TILEB + 16 * ks * W::ROWB;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(uintptr_t)a0);
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (s16x4 __attribute__((address_space(3)))*)(uintptr_t)(a0 + 8 * W::ROWB));
                const bf16x8 av = __builtin_bit_cast(bf16x8, s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
#pragma unroll
                for (int r = 0; r < RT; ++r)
                    if (act[r]) o[r][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, pf[r][ks], o[r][dt], 0, 0, 0);
            }
    };
    auto pair_step = [&](const int k, const int pb) {              // tiles 2k, 2k+1 in buffers 2pb, 2pb+1 (pb static)
        wait_vmcnt<0>();
        __syncthreads();
        if (2 * k + 2 < n_tiles) issue(2 * k + 2);
        if (2 * k + 3 < n_tiles) issue(2 * k + 3);
        const int base_a = 64 * k, base_b = 64 * k + 32;
        bool act_a[RT], act_b[RT], any_a = false, any_b = false;   // wave-uniform: row tile not past its diagonal
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            act_a[r] = base_a < kv_end[r];
            act_b[r] = 2 * k + 1 < n_tiles && base_b < kv_end[r];
            any_a |= act_a[r]; any_b |= act_b[r];
        }
        if (!any_a) return;
        bf16x8 kfa[W::KS], kfb[W::KS], pfa[RT][2], pfb[RT][2];
        f32x16 sa[RT], sb[RT];
        read_k(kfa, 2 * pb);                                       // K fragments are read once and shared by the row tiles
#pragma unroll
        for (int r = 0; r < RT; ++r)
            if (act_a[r]) qk(sa[r], kfa, r);
        if (any_b) {
            read_k(kfb, 2 * pb + 1);
#pragma unroll
            for (int r = 0; r < RT; ++r)
                if (act_b[r]) qk(sb[r], kfb, r);
        }
#pragma unroll
        for (int r = 0; r < RT; ++r)
            if (act_a[r]) softmax(sa[r], pfa[r], r, base_a);
        pv(pfa, act_a, 2 * pb);
        if (any_b) {
#pragma unroll
            for (int r = 0; r < RT; ++r)
                if (act_b[r]) softmax(sb[r], pfb[r], r, base_b);
            pv(pfb, act_b, 2 * pb + 1);
        }
    };
    for (int k0 = 0; 2 * k0 < n_tiles; k0 += 2) {
        pair_step(k0, 0);
        if (2 * (k0 + 1) < n_tiles) pair_step(k0 + 1, 1);
    }

#pragma unroll
    for (int r = 0; r < RT; ++r) {
        float la, lb;
        lane_pair<32>(l_run[r], la, lb);
        const float inv = 1.f / (la + lb);
        if (valid[r]) {
            uint16_t* op = p.out + (qrow0 + jrow[r]) * p.o_stride + (int64_t)(h * G + grow[r]) * D + 4 * hh;
#pragma unroll
            for (int dt = 0; dt < W::DT; ++dt)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    uint2 w;
                    w.x = pack_bf16(o[r][dt][4 * q4 + 0] * inv, o[r][dt][4 * q4 + 1] * inv);
                    w.y = pack_bf16(o[r][dt][4 * q4 + 2] * inv, o[r][dt][4 * q4 + 3] * inv);
                    *reinterpret_cast<uint2*>(op + 32 * dt + 8 * q4) = w;
                }
        }
    }
}

static int ilog2_exact_p(int32_t x) {
    if (x <= 0 || (x & (x - 1))) return -1;
    int s = 0;
    while ((1 << s) < x) ++s;
    return s;
}

static int prefill_mode() {      // tuning / cross-check hook: 1 = row-parallel (VALU) path, 2 = one-wave-per-tile MFMA kernel
    static int v = [] { const char* e = getenv("M355_PREFILL_MODE"); return e ? atoi(e) : 0; }();
    return v;
}

static int prefill_rt() {        // row tiles per wave of the workgroup-shared kernel (1 or 2)
    static int v = [] { const char* e = getenv("M355_PREFILL_RT"); const int x = e ? atoi(e) : 1; return x == 2 ? 2 : 1; }();
    return v;
}

template <int D, int G>
static void launch_prefill_wg(PrefillParams p, int max_extend_len, hipStream_t st) {
    const int rt = prefill_rt();
    p.tiles_max = (max_extend_len * G + 128 * rt - 1) / (128 * rt);
    const dim3 grid((unsigned)((int64_t)p.B * p.Hkv * p.tiles_max)), block(256);
    constexpr int NBUF = D == 64 ? 4 : 4;
    const size_t lds = NBUF * WgCfg<D>::BUFB + kSlotWin * 4;   // K tiles, V tiles, slot window
    static bool attr_done = [] {
        (void)hipFuncSetAttribute((const void*)prefill_wg_kernel<D, G, 1, NBUF, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)prefill_wg_kernel<D, G, 1, NBUF, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)prefill_wg_kernel<D, G, 2, NBUF, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)prefill_wg_kernel<D, G, 2, NBUF, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        return true; }();
    (void)attr_done;
#define M355_WG(RR, PP) hipLaunchKernelGGL((prefill_wg_kernel<D, G, RR, NBUF, PP>), grid, block, lds, st, p)
    if (rt == 1) { if (p.ps_shift >= 0) M355_WG(1, true); else M355_WG(1, false); }
    else { if (p.ps_shift >= 0) M355_WG(2, true); else M355_WG(2, false); }
#undef M355_WG
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
#define M355_CASE(DD, GG) if (D == DD && G == GG) { \
        if (prefill_mode() == 2) launch_prefill<DD, GG>(p, st); else launch_prefill_wg<DD, GG>(p, max_extend_len, st); \
        return check_launch(); }
    M355_CASE(64, 1) M355_CASE(64, 2) M355_CASE(64, 4) M355_CASE(64, 8)
    M355_CASE(128, 1) M355_CASE(128, 2) M355_CASE(128, 4) M355_CASE(128, 8)
#undef M355_CASE
    return M355_ERR_UNSUPPORTED;
}
