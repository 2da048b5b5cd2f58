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

#include "prefill_common.h"

namespace m355 {

#ifdef M355_DEV
static unsigned long long* g_prefill_stamps = nullptr;      // dev build: timing probes (tools/stamp_prefill.py)
#endif

#ifdef M355_DEV      // ---- one wave per row tile, K/V straight from L2: an A/B reference (M355_PREFILL_MODE=2), dev build only
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

#endif  // M355_DEV

// ------------------------------------------------------------------------------------------------
// Workgroup-shared variant (default): the 4 waves of a workgroup own 4 consecutive row tiles of one
// (request, kv head) and walk the request's KV together.  Every 32-token K/V tile is brought into LDS ONCE per
// workgroup with global_load_lds_dwordx4 (whole 2*D-byte rows, XOR-swizzled through the choice of the chunk each
// lane fetches, double buffered) and consumed by all waves: K fragments are ds_read_b128 rows, V^T fragments
// ds_read_b64_tr_b16 transposed reads.  L2->CU traffic drops by 4 against one-wave-per-tile streaming, which was
// what bounded the first kernel (3.3 GB per layer on the 8B shared-prefix workload).
// ------------------------------------------------------------------------------------------------
template <int D, int G, int NW, int NBUF, bool POW2, bool STAMP = false, bool F16 = false>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void prefill_wg_kernel(const PrefillParams p) {
    using W = WgCfg<D>;
    constexpr int NT = 64 * NW;                       // threads
    constexpr int IPW = 2 * W::NI / NW;               // DMA instructions per wave per KV tile (K and V)
    constexpr int NP = NBUF / 2;                      // pair slots of the tile ring; NP-1 pairs are in flight ahead
    static_assert(NBUF % 2 == 0 && NP >= 2 && (2 * W::NI) % NW == 0, "ring geometry");
    extern __shared__ __attribute__((aligned(1024))) uint8_t smem[];     // NBUF K tiles, NBUF V tiles, the slot window
    int32_t* const slotw = reinterpret_cast<int32_t*>(smem + NBUF * W::BUFB);
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int l = threadIdx.x & 63;
    const int c = l & 31, hh = l >> 5;
    constexpr int ROWS_WG = 32 * NW;
    static_assert(W::KS == 2 * W::DT, "as many K fragments as V^T fragments per tile: they share registers");

    // XCD-aware order: workgroups go round-robin over the 8 XCDs, so blockIdx % 8 names an XCD.  All row blocks of one
    // (request, kv head) - which stream the same K/V - are given to ONE XCD, next to each other in its dispatch
    // order and longest (latest) first, so that its L2 fetches that K/V once instead of all eight L2s fetching it
    // (L2 hit rate 64 % -> 90 % on the shared-prefix workload).  Pairs left over after the last full round of 8 are
    // spread over all XCDs as before (one pair on one XCD would leave seven idle).
    unsigned long long r_entry = 0;
    if constexpr (STAMP) r_entry = __builtin_amdgcn_s_memrealtime();
    const int n_bh = p.B * p.Hkv;
    const int mapped = (n_bh >> 3) * 8 * p.tiles_max;              // workgroups of the full rounds
    int tb, bh;
    if ((int)blockIdx.x < mapped) {
        int item = blockIdx.x >> 3;
        // ... except each pair's EARLIEST row block (the shortest walk): those come after all the others, so the launch's
        // last, partly filled round is made of the shortest blocks (cfg3 144.0 -> 142.3 us, twice the groups 277 -> 270;
        // holding back two or four per pair is no better: profiles/r03_prefill_order_ab.txt)
        const int hold = p.tiles_max > 1 ? 1 : 0, lead = p.tiles_max - hold;
        const int n_lead = (n_bh >> 3) * lead;
        if (item < n_lead) { tb = p.tiles_max - 1 - item % lead; bh = (item / lead) * 8 + (blockIdx.x & 7); }
        else { tb = 0; bh = (item - n_lead) * 8 + (blockIdx.x & 7); }
    } else {
        const int item = blockIdx.x - mapped;
        tb = p.tiles_max - 1 - item % p.tiles_max;
        bh = (n_bh & ~7) + item / p.tiles_max;
    }
    const int h = bh % p.Hkv;
    const int b = bh / p.Hkv;
    const int P = (int)p.prefix_lens[b];
    const int E = (int)p.seq_lens[b] - P;
    const int rows_b = E * G;
    if (ROWS_WG * tb >= rows_b) return;                            // whole workgroup: no barrier has been reached yet
    const int32_t* pt_row = p.page_table + p.page_table_ids[b] * p.pt_stride;
    const int64_t qrow0 = p.cu_q[b];
    const int j_last_wg = min(E - 1, (ROWS_WG * tb + ROWS_WG - 1) / G);
    const int kv_end_wg = P + j_last_wg + 1;                       // tokens any row of the workgroup attends
    const int n_tiles = (kv_end_wg + 31) >> 5;
    const int n_pairs_all = (n_tiles + 1) >> 1;
    // split-KV (few row blocks, long KV): this workgroup walks pairs [p_begin, n_pairs) only and leaves fp32 partials
    const int p_begin = p.splits > 1 ? (int)((int64_t)n_pairs_all * blockIdx.y / p.splits) : 0;
    const int n_pairs = p.splits > 1 ? (int)((int64_t)n_pairs_all * (blockIdx.y + 1) / p.splits) : n_pairs_all;   // exclusive end

    // ---- this wave's row tile: the lane's row, its limit, Q^T fragments
    const int t = tb * NW + wid;                                   // row tile index inside the request
    const int mrow = 32 * t + c;
    const bool valid = mrow < rows_b;
    const int jrow = mrow / G, grow = mrow % G;
    const int limit = valid ? P + jrow + 1 : 0;
    const int kv_end = 32 * t < rows_b ? P + min(E - 1, (32 * t + 31) / G) + 1 : 0;   // wave-uniform
    const int kv_min = P + (32 * t) / G + 1;
    bf16x8 qf[W::KS];
    f32x16 o[W::DT];
    {
        const uint16_t* qp = p.q + (qrow0 + jrow) * p.q_stride + (int64_t)(h * G + grow) * D + hh * 8;
#pragma unroll
        for (int s = 0; s < W::KS; ++s) {
            u32x4 v = u32x4{0u, 0u, 0u, 0u};
            if (valid) v = *reinterpret_cast<const u32x4*>(qp + s * 16);          // d = 16 s + 8 hh + j: the operand's own order
            qf[s] = __builtin_bit_cast(bf16x8, v);
        }
#pragma unroll
        for (int dt = 0; dt < W::DT; ++dt)
#pragma unroll
            for (int i = 0; i < 16; ++i) o[dt][i] = 0.f;
    }
    // The running max starts at a floor far below any real score but far ABOVE a masked one (kNegBig * scale): a row
    // whose first pair is entirely masked - possible in a later KV split - then keeps the floor and gets p = 0 for
    // every masked entry, instead of "max = the mask value" and p = 2^(rounding noise of 1e29).
    constexpr float kRunFloor = -1.0e20f;
    float m_run = kRunFloor, l_run = 0.f;

    // ---- DMA duty of this wave: instruction ids wid, wid + 4, ... of the tile's 2*NI (K first, then V)
    const int drow = l / W::C, dch = l % W::C;
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)smem);
    const uint8_t* kbase = reinterpret_cast<const uint8_t*>(p.k_pool) + (int64_t)h * W::ROWB;
    const uint8_t* vbase = reinterpret_cast<const uint8_t*>(p.v_pool) + (int64_t)h * W::ROWB;
    const uint32_t row_bytes = (uint32_t)(p.row_elems * 2);       // (< 2^32: a token row is at most a few KiB)
    // slots of tokens [win*kSlotWin, +kSlotWin) -> LDS (ordinary loads: the compiler's wait drains the DMA queue too,
    // once per 32 KV tiles)
    int cur_win = -1;
    auto fill_slots = [&](int win) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kSlotWin / NT; ++k) {
            int tok = win * kSlotWin + (int)threadIdx.x + NT * k;
            tok = tok < kv_end_wg ? tok : kv_end_wg - 1;
            const int pg = POW2 ? (tok >> p.ps_shift) : (tok / p.ps);
            slotw[threadIdx.x + NT * k] = pt_row[pg] * p.ps + (tok - (POW2 ? (pg << p.ps_shift) : pg * p.ps));
        }
        __syncthreads();
        cur_win = win;
    };
    // This wave's share of KV tiles 2*pair, 2*pair+1 (ring buffers buf0, buf0+1): IPW K pieces and IPW V pieces of
    // 1 KiB, K and V of the same token rows, so one slot look-up serves a K piece and a V piece.  A pair or tile index
    // past the end re-fetches the last one (finite bytes that the mask removes, into a vacated buffer), so every
    // pair costs every wave the same number of DMA instructions and the vmcnt arithmetic is static.
    // Piece x (0..IPW-1): tile x / (IPW/2) of the pair, DMA row block i = wid + NW * (x % (IPW/2)).
    static_assert(IPW % 2 == 0 && W::NI % NW == 0, "a wave's K pieces and V pieces cover the same rows");
    constexpr int HP = IPW / 2;
    auto pair_slots = [&](int pair, int32_t (&sl)[IPW]) {
        pair = pair < n_pairs ? pair : n_pairs - 1;
        pair = pair > p_begin ? pair : p_begin;            // (an empty split: clamped to a pair that exists)
        pair = pair < n_pairs_all ? pair : n_pairs_all - 1;
        if ((2 * pair) / (kSlotWin / 32) != cur_win) fill_slots((2 * pair) / (kSlotWin / 32));   // workgroup-uniform
#pragma unroll
        for (int x = 0; x < IPW; ++x) {
            const int tile = 2 * pair + x / HP < n_tiles ? 2 * pair + x / HP : n_tiles - 1;
            sl[x] = slotw[((tile * 32) & (kSlotWin - 1)) + (wid + NW * (x % HP)) * W::RPI + drow];
        }
    };
    // per-lane byte offsets of the 16-byte chunk inside a token row (K and V swizzles), one per DMA row block of this wave
    uint32_t kch[HP], vch[HP];
#pragma unroll
    for (int y = 0; y < HP; ++y) {
        const int row = (wid + NW * y) * W::RPI + drow;
        kch[y] = 16 * (dch ^ W::sK(row));
        vch[y] = 16 * (dch ^ W::sV(row));
    }
    auto piece = [&](const bool is_v, const int x, const int32_t slot, const int buf0) {
        const int i = wid + NW * (x % HP);                         // wave-uniform
        // slot * row_bytes + chunk: ONE v_mad_u64_u32 (32 x 32 -> 64 bits), not a 64 x 64 product
        const uint64_t off = (uint64_t)(uint32_t)slot * row_bytes + (is_v ? vch[x % HP] : kch[x % HP]);
        glds16p<M355_PREFILL_POLICY>((is_v ? vbase : kbase) + off,
               lds0 + (is_v ? NBUF * W::TILEB : 0) + (buf0 + x / HP) * W::TILEB + 1024 * i);
    };
    auto issue_pair = [&](const int pair, const int buf0) {         // prologue: everything at once
        int32_t sl[IPW];
        pair_slots(pair, sl);
#pragma unroll
        for (int x = 0; x < IPW; ++x) { piece(false, x, sl[x], buf0); piece(true, x, sl[x], buf0); }
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
    static_assert(2 * NBUF * W::TILEB <= 131072, "ring fits the LDS");

    // the first slot window is fetched beside the q fragments (one memory latency for both; its loads are waited for
    // inside fill_slots, the q fragments - issued earlier, returned in order - with them)
    fill_slots((2 * p_begin) / (kSlotWin / 32));
    wait_vmcnt<0>();                                               // q fragments are in: the DMA count starts clean
#pragma unroll
    for (int j = 0; j < NP - 1; ++j)
        issue_pair(p_begin + j, 2 * j);

    // K fragment s of the tile in buffer `buf` (ds_read_b128) and V^T fragment f = 2*dt + ks (two transposed reads:
    // token rows 16 ks + 4 hh + q and + 8; the swizzle only depends on the row modulo 8)
    auto rdk = [&](const int buf, const int s) {
        return *reinterpret_cast<const bf16x8 __attribute__((address_space(3)))*>((uintptr_t)(kaddr[s] + buf * W::TILEB));
    };
    auto rdv = [&](const int buf, const int f) {
        const uint32_t a0 = vaddr[f >> 1] + buf * W::TILEB + 16 * (f & 1) * W::ROWB;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(uintptr_t)a0);
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(uintptr_t)(a0 + 8 * W::ROWB));
        return __builtin_bit_cast(bf16x8, s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
    };
    auto mask = [&](f32x16& s, const int base) {                   // lane holds tokens base + (i&3) + 8 (i>>2) + 4 hh
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int tok = base + (i & 3) + 8 * (i >> 2) + 4 * hh;
            s[i] = tok < limit ? s[i] : kNegBig;
        }
    };
    auto tile_max = [&](const f32x16& s) {
        float mx = s[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) mx = fmaxf(mx, s[i]);
        return mx;
    };
    // lazy running max: it moves only when beaten by more than 2^kLazy, so the rescale of the O^T accumulators is
    // rare; p <= 2^kLazy stays well inside bf16 / fp32 range.  Returns alpha and rescales O^T when it is not 1.
    auto new_max = [&](const float mx) {
        float ma, mb;
        lane_pair<32>(mx, ma, mb);
        const float m_tile = fmaxf(ma, mb) * p.scale_log2;         // scores stay raw; scale > 0 commutes with max
        const float m_new = m_tile > m_run + kLazy ? m_tile : m_run;
        const float alpha = exp2_fast(m_run - m_new);
        m_run = m_new;
        if (__any(alpha != 1.f)) {
#pragma unroll
            for (int dt = 0; dt < W::DT; ++dt)
#pragma unroll
                for (int i = 0; i < 16; ++i) o[dt][i] *= alpha;
        }
        return alpha;
    };
    auto exp_pack = [&](f32x16& s, bf16x8 (&pf)[2]) {              // p = 2^(s*scale - m); returns the lane's sum
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) { s[i] = exp2_fast(fmaf(s[i], p.scale_log2, -m_run)); sum += s[i]; }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            u32x4 w;
#pragma unroll
            for (int x = 0; x < 4; ++x) w[x] = pack16<F16>(s[8 * ks + 2 * x], s[8 * ks + 2 * x + 1]);
            pf[ks] = __builtin_bit_cast(bf16x8, w);
        }
        return sum;
    };

    // KV tiles are consumed in PAIRS (tiles 2k, 2k+1 in buffers 2pb, 2pb+1; pb static).  Per pair: wait for this
    // wave's DMA, barrier (both tiles landed; everyone left the previous pair), send the next pair into the buffers
    // the previous pair vacated, then
    //   all 16 K fragments are requested at once; S_a^T, S_b^T = K Q^T (16 MFMAs; the V^T fragments of tile a are
    //   read into tile a's K registers behind the second eight, tile b's after them),
    //   ONE max / rescale decision for the 64 tokens (the only branches), then a straight line of
    //   exp2 + pack (VALU) and O^T += V^T P^T (16 MFMAs) that the scheduler interleaves.
    // No MFMA waits on an LDS read it has just issued, and the partner wave on the SIMD finds the matrix pipe free
    // while this one is in its VALU stretch.
    unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0}, tprev = 0, tfirst = 0, rfirst = 0;
    if constexpr (STAMP) { tprev = tfirst = __builtin_amdgcn_s_memtime(); rfirst = __builtin_amdgcn_s_memrealtime(); }
    auto stamp = [&](const int i) {
        if constexpr (STAMP) { const unsigned long long t = __builtin_amdgcn_s_memtime(); tacc[i] += t - tprev; tprev = t; }
    };
    auto pair_step = [&](const int k, auto pb_const) {
        constexpr int pb = decltype(pb_const)::value;
        stamp(4);
        wait_vmcnt<(NP - 2) * 2 * IPW>();                          // pair k has landed; NP-2 younger pairs stay in flight
        __syncthreads();
        stamp(0);
        // the pair NP-1 ahead goes into the buffers that pair k-1 vacated (slots now, DMA further down)
        constexpr int nb = 2 * ((pb + NP - 1) % NP);
        int32_t sl[IPW];
        const int base_a = 64 * k, base_b = 64 * k + 32;
        if (base_a >= kv_end) {                                    // wave-uniform: the row tile is past its diagonal
            pair_slots(k + NP - 1, sl);
#pragma unroll
            for (int x = 0; x < IPW; ++x) { piece(false, x, sl[x], nb); piece(true, x, sl[x], nb); }
            return;
        }
        bf16x8 fa[W::KS], fb[W::KS], pa[2], pbk[2];
        f32x16 sa, sb;
#pragma unroll
        for (int s = 0; s < W::KS; ++s) fa[s] = rdk(2 * pb, s);
        // the slot look-ups queue up behind the K fragments: their LDS latency passes under the QK^T MFMAs instead of in
        // front of them (they are first needed by the K pieces below)
        pair_slots(k + NP - 1, sl);
        stamp(1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 16; ++i) { sa[i] = 0.f; sb[i] = 0.f; }
        if (kPrio) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < W::KS; ++s) {                          // S_a^T; tile b's K fragments fill in behind
            sa = mfma_32x32x16<F16>(fa[s], qf[s], sa);
            fb[s] = rdk(2 * pb + 1, s);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < W::KS; ++s) {                          // S_b^T; tile a's V^T fragments take tile a's K registers
            sb = mfma_32x32x16<F16>(fb[s], qf[s], sb);
            fa[s] = rdv(2 * pb, s);
        }
        if (kPrio) __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        stamp(2);
        // the next pair's DMA goes out in the VALU-only stretches (beside LDS reads and MFMA issue a piece costs the
        // wave several times more): K pieces before the max, V pieces before the exponentials
#pragma unroll
        for (int x = 0; x < IPW; ++x) piece(false, x, sl[x], nb);
        // (a tile b that does not exist - its buffer then holds the last tile again - or lies past this wave's
        // diagonal is fully masked and adds exactly 0)
        if (base_b + 32 > kv_min) { mask(sa, base_a); mask(sb, base_b); }         // wave-uniform: only diagonal pairs
        const float alpha = new_max(fmaxf(tile_max(sa), tile_max(sb)));
        stamp(3);
#pragma unroll
        for (int x = 0; x < IPW; ++x) piece(true, x, sl[x], nb);
        float sum = exp_pack(sa, pa);
        __builtin_amdgcn_sched_barrier(0);
        if (kPrio == 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < W::KS; ++s) {                          // O^T += V_a^T P_a^T; tile b's V^T fragments fill in behind
            o[s >> 1] = mfma_32x32x16<F16>(fa[s], pa[s & 1], o[s >> 1]);
            fb[s] = rdv(2 * pb + 1, s);
        }
        sum += exp_pack(sb, pbk);                                  // independent of the MFMAs above: interleaved with them
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < W::KS; ++s) o[s >> 1] = mfma_32x32x16<F16>(fb[s], pbk[s & 1], o[s >> 1]);
        if (kPrio == 2) __builtin_amdgcn_s_setprio(0);
        l_run = fmaf(l_run, alpha, sum);
        stamp(4);
    };
    for (int k0 = p_begin; k0 < n_pairs; k0 += NP) {
        pair_step(k0, IntC<0>{});
        if (k0 + 1 < n_pairs) pair_step(k0 + 1, IntC<1>{});
        if constexpr (NP > 2) { if (k0 + 2 < n_pairs) pair_step(k0 + 2, IntC<2>{}); }
        if constexpr (NP > 3) { if (k0 + 3 < n_pairs) pair_step(k0 + 3, IntC<3>{}); }
        static_assert(NP <= 4, "unrolled pair slots");
    }

    wait_vmcnt<0>();                                               // the clamped re-fetches of the last pairs must not outlive the wave
    if constexpr (STAMP) {
        if (l == 0 && p.stamps) {
            unsigned long long* d = p.stamps + ((int64_t)blockIdx.x * NW + wid) * 16;
            for (int i = 0; i < 5; ++i) d[i] = tacc[i];
            d[8] = r_entry; d[9] = rfirst; d[10] = __builtin_amdgcn_s_memrealtime();   // absolute: kernel entry, first pair, loop end
            d[11] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15;                      // XCC_ID
            d[5] = __builtin_amdgcn_s_memrealtime() - rfirst;      // 100 MHz ticks (the 'between pairs' sum is folded into phase 4)
            d[6] = __builtin_amdgcn_s_memtime() - tfirst;
            d[7] = (unsigned long long)n_pairs;
        }
    }
    float la, lb;
    lane_pair<32>(l_run, la, lb);
    if (p.splits > 1) {                                            // unnormalised partial of this split: O^T, m, l
        if (valid) {
            float* wo = p.ws + (((qrow0 + jrow) * p.Hq + (h * G + grow)) * p.splits + blockIdx.y) * (D + 4);
#pragma unroll
            for (int dt = 0; dt < W::DT; ++dt)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4)
                    *reinterpret_cast<f32x4*>(wo + 32 * dt + 8 * q4 + 4 * hh) =
                        f32x4{o[dt][4 * q4 + 0], o[dt][4 * q4 + 1], o[dt][4 * q4 + 2], o[dt][4 * q4 + 3]};
            if (hh == 0) { wo[D] = m_run; wo[D + 1] = la + lb; }
        }
        return;
    }
    const float inv = 1.f / (la + lb);
    if (valid) {
        uint16_t* op = p.out + (qrow0 + jrow) * p.o_stride + (int64_t)(h * G + grow) * D + 4 * hh;
#pragma unroll
        for (int dt = 0; dt < W::DT; ++dt)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                uint2 w;
                w.x = pack16<F16>(o[dt][4 * q4 + 0] * inv, o[dt][4 * q4 + 1] * inv);
                w.y = pack16<F16>(o[dt][4 * q4 + 2] * inv, o[dt][4 * q4 + 3] * inv);
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

static int prefill_mode() {      // dev build cross-check hook: 2 = the older one-wave-per-tile MFMA kernel
    static const int v = dev_env_int("M355_PREFILL_MODE", 0);
    return v;
}

// KV splits per row block: only when the whole launch has too few row blocks (T*Hq/128 of them, all requests and
// heads) to occupy the chip's 512 workgroup slots and the KV is long - a short new turn over a long cached history.
// Splits bring the workgroup count to about 512, each keeping at least 512 tokens; the partials stay small
// (splits * T * Hq <= 2^16 rows of D + 4 floats).  The same rule sizes the workspace.  (With 270 blocks - every CU
// already busy - 4 splits measured no gain: 260 us either way; with 15 blocks 8 splits gave 319 -> 60 us.)
static int prefill_splits(int64_t T, int32_t Hq, int32_t max_seq_len) {
    static const int forced = dev_env_int("M355_PREFILL_SPLITS", 0);   // dev build A/B hook: 1 = never, n > 1 = n splits
    if (forced >= 1) return forced > 32 ? 32 : forced;
    const int64_t blocks = (T * Hq + 127) / 128;
    int s = (int)(512 / (blocks > 0 ? blocks : 1));
    if (s > 32) s = 32;
    while (s > 1 && max_seq_len / s < 512) --s;
    // ... and rather 1 024 while that still leaves 256 workgroups: at 8 pairs per workgroup the ramp and twice the partials
    // to merge cost more than the second workgroup per CU brings (2 x 32 tokens over 16 384 cached: 58.8 -> 49.8 us)
    while (s > 1 && max_seq_len / s < 1024 && blocks * (s - 1) >= 256) --s;
    return s < 1 ? 1 : s;
}

// out[row, head, :] = sum_s O_s 2^(m_s - M) / sum_s l_s 2^(m_s - M): 32 lanes per (row, head), 4 floats each (D = 128)
// or 16 lanes (D = 64)
template <int D, bool F16 = false>
__global__ __launch_bounds__(256) void prefill_merge_kernel(const float* __restrict__ ws, uint16_t* __restrict__ out,
                                                            int64_t o_stride, int64_t n_items, int32_t Hq, int32_t S) {
    constexpr int LPI = D / 4;                                      // lanes per item
    const int64_t item = ((int64_t)blockIdx.x * 256 + threadIdx.x) / LPI;
    const int li = threadIdx.x % LPI;
    if (item >= n_items) return;
    const float* base = ws + item * S * (D + 4);
    // one pass, eight splits at a time: their (m, l) pairs and 16-byte pieces are all requested before the first is used
    // (the kernel is a few waves per CU: with one load in flight per lane it was pure latency, 9.3 us for 17 MB)
    constexpr int CH = 8;
    float M = kNegBig, den = 0.f;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int s0 = 0; s0 < S; s0 += CH) {
        float m[CH], l[CH];
        f32x4 o[CH];
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const float* ps = base + (int64_t)(s0 + k < S ? s0 + k : S - 1) * (D + 4);
            const float2 ml = *reinterpret_cast<const float2*>(ps + D);
            m[k] = s0 + k < S ? ml.x : kNegBig;
            l[k] = ml.y;
            o[k] = *reinterpret_cast<const f32x4*>(ps + 4 * li);
        }
        float Mn = M;
#pragma unroll
        for (int k = 0; k < CH; ++k) Mn = fmaxf(Mn, m[k]);
        const float r = exp2_fast(M - Mn);                           // (first chunk: 2^(-1e30 - Mn) = 0 on zeros)
        den *= r;
        acc *= r;
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const float w = s0 + k < S ? exp2_fast(m[k] - Mn) : 0.f;
            den = fmaf(w, l[k], den);
            acc += o[k] * w;
        }
        M = Mn;
    }
    const float inv = 1.f / den;
    const int64_t row = item / Hq, head = item % Hq;
    uint2 w;
    w.x = pack16<F16>(acc[0] * inv, acc[1] * inv);
    w.y = pack16<F16>(acc[2] * inv, acc[3] * inv);
    *reinterpret_cast<uint2*>(out + row * o_stride + head * D + 4 * li) = w;
}

static int prefill_wg_cfg() {    // tuning hook: M355_PREFILL_WG = 10*waves + ring tiles/4, e.g. 41 (4 waves, 4 tiles), 42, 82
    static const int v = dev_env_int("M355_PREFILL_WG", 0);
    return v;
}

template <int D, int G, int NW, int NBUF>
static void launch_prefill_wg_cfg(PrefillParams p, int max_extend_len, hipStream_t st) {
    p.tiles_max = (max_extend_len * G + 32 * NW - 1) / (32 * NW);
    const dim3 grid((unsigned)((int64_t)p.B * p.Hkv * p.tiles_max), (unsigned)p.splits), block(64 * NW);
    const size_t lds = NBUF * WgCfg<D>::BUFB + kSlotWin * 4;   // K tiles, V tiles, slot window
    static bool attr_done = [] {
        (void)hipFuncSetAttribute((const void*)prefill_wg_kernel<D, G, NW, NBUF, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)prefill_wg_kernel<D, G, NW, NBUF, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        return true; }();
    (void)attr_done;
#ifdef M355_DEV
    if constexpr (D == 128 && G == 4) {                            // phase stamps (tools/stamp_prefill.py)
        if (g_prefill_stamps && p.ps_shift >= 0) {
            p.stamps = g_prefill_stamps;
            (void)hipFuncSetAttribute((const void*)prefill_wg_kernel<D, G, NW, NBUF, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            hipLaunchKernelGGL((prefill_wg_kernel<D, G, NW, NBUF, true, true>), grid, block, lds, st, p);
            return;
        }
    }
#endif
    if (p.f16) {                                                   // (default geometry only: NW = 4, 4 tile buffers)
        if constexpr (NW == 4 && NBUF == 4) {
            static bool attr16 = [] {
                (void)hipFuncSetAttribute((const void*)prefill_wg_kernel<D, G, 4, 4, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                (void)hipFuncSetAttribute((const void*)prefill_wg_kernel<D, G, 4, 4, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                return true; }();
            (void)attr16;
            if (p.ps_shift >= 0) hipLaunchKernelGGL((prefill_wg_kernel<D, G, 4, 4, true, false, true>), grid, block, lds, st, p);
            else hipLaunchKernelGGL((prefill_wg_kernel<D, G, 4, 4, false, false, true>), grid, block, lds, st, p);
        }
        return;
    }
    if (p.ps_shift >= 0) hipLaunchKernelGGL((prefill_wg_kernel<D, G, NW, NBUF, true>), grid, block, lds, st, p);
    else hipLaunchKernelGGL((prefill_wg_kernel<D, G, NW, NBUF, false>), grid, block, lds, st, p);
}

#ifdef M355_DEV
// The dev build can route head_dim 128 to the 64-rows-per-wave kernel of prefill_wg2.hip (M355_PREFILL_RT=2).  It is at
// parity with prefill_wg_kernel, not ahead, so the product has one kernel and does not link the other.
static bool prefill_use_wg2(int D) { return D == 128 && dev_env_int("M355_PREFILL_RT", 1) == 2; }
#endif

template <int D, int G>
static void launch_prefill_wg(const PrefillParams& p, int max_extend_len, hipStream_t st) {
#ifdef M355_DEV
    const int cfg = p.f16 ? 0 : prefill_wg_cfg();
    if constexpr (D == 64) {
        if (cfg == 42) return launch_prefill_wg_cfg<D, G, 4, 8>(p, max_extend_len, st);
    } else {
        if (cfg == 82) return launch_prefill_wg_cfg<D, G, 8, 8>(p, max_extend_len, st);
    }
#endif
    return launch_prefill_wg_cfg<D, G, 4, 4>(p, max_extend_len, st);
}

#ifdef M355_DEV
template <int D, int G>
static void launch_prefill(const PrefillParams& p, hipStream_t st) {
    const int64_t items = (int64_t)p.B * p.Hkv * p.tiles_max;
    const dim3 grid((unsigned)((items + 3) / 4)), block(256);
    const size_t lds = 4 * PrefillCfg<D>::VTILE;
    if (p.ps_shift >= 0) hipLaunchKernelGGL((prefill_mfma_kernel<D, G, true>), grid, block, lds, st, p);
    else hipLaunchKernelGGL((prefill_mfma_kernel<D, G, false>), grid, block, lds, st, p);
}
#endif

}  // namespace m355

using namespace m355;

#ifdef M355_DEV
extern "C" int m355dbg_set_prefill_stamp_buffer(void* dev_ptr) {   // dev build only (not part of include/mi355attn.h)
    m355::g_prefill_stamps = (unsigned long long*)dev_ptr;
    return 0;
}
#endif

extern "C" int64_t m355_prefill_workspace_bytes(int32_t T, int32_t Hq, int32_t D, int32_t max_seq_len_hint) {
    if (T <= 0 || Hq <= 0 || D <= 0 || max_seq_len_hint <= 0) return 0;
    if (prefill_mode() == 0) {                            // split-KV partials of the workgroup kernel (few rows, long KV)
        const int sp = prefill_splits(T, Hq, max_seq_len_hint);
        return sp > 1 ? (int64_t)T * Hq * sp * (D + 4) * (int64_t)sizeof(float) : 0;
    }
    return 0;
}

extern "C" int m355_prefill_attn(const void* q, int64_t q_row_stride, const void* k_pool, const void* v_pool,
                                 int64_t pool_row_stride, const int32_t* page_table, int64_t pt_row_stride, const int64_t* page_table_ids,
                                 const int64_t* seq_lens, const int64_t* prefix_lens, const int32_t* cu_q, void* out,
                                 int64_t o_row_stride, int32_t B, int32_t T, int32_t max_extend_len,
                                 int32_t max_seq_len_hint, int32_t Hq, int32_t Hkv, int32_t D, int32_t page_size,
                                 float scale, void* workspace, int64_t workspace_bytes, int32_t dtype, void* stream) {
    if (T == 0 || B == 0) return M355_OK;
    if (!q || !k_pool || !v_pool || !page_table || !out || !seq_lens || !prefix_lens || !cu_q || !page_table_ids || B < 0 ||
        T < 0 || max_extend_len <= 0 || Hq <= 0 || Hkv <= 0 || Hq % Hkv || page_size <= 0)
        return M355_ERR_INVALID_ARG;
    if (pool_row_stride == 0) pool_row_stride = (int64_t)Hkv * D;
    if (pool_row_stride < (int64_t)Hkv * D || pool_row_stride % 8) return M355_ERR_INVALID_ARG;
    const bool f16 = dtype == M355_DTYPE_F16;          // IEEE half: the workgroup kernel only
    if (dtype != M355_DTYPE_BF16 && !(f16 && prefill_mode() == 0)) return M355_ERR_UNSUPPORTED;
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
    p.q_stride = q_row_stride; p.o_stride = o_row_stride; p.pt_stride = pt_row_stride; p.row_elems = pool_row_stride;
    p.B = B; p.Hq = Hq; p.Hkv = Hkv; p.ps = page_size; p.ps_shift = ilog2_exact_p(page_size);
    p.tiles_max = (max_extend_len * G + 31) / 32;
    p.scale_log2 = scale * kLog2e;
    p.stamps = nullptr;
    p.ws = nullptr;
    p.splits = 1;
    p.f16 = f16 ? 1 : 0;
    if (prefill_mode() == 0) {
        p.splits = prefill_splits(T, Hq, max_seq_len_hint);
        if (p.splits > 1) {
            const int64_t need = (int64_t)T * Hq * p.splits * (D + 4) * (int64_t)sizeof(float);
            if (!workspace || workspace_bytes < need || ((uintptr_t)workspace & 15)) return M355_ERR_WORKSPACE;
            p.ws = (float*)workspace;
        }
    }
    if ((int64_t)B * Hkv * p.tiles_max > 0x7fffffffLL) return M355_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
#ifdef M355_DEV
#define M355_PREFILL_LAUNCH(DD, GG) if (prefill_mode() == 2) launch_prefill<DD, GG>(p, st); \
        else if (!(prefill_use_wg2(DD) && launch_prefill_wg2(p, DD, GG, max_extend_len, st))) launch_prefill_wg<DD, GG>(p, max_extend_len, st);
#else
#define M355_PREFILL_LAUNCH(DD, GG) launch_prefill_wg<DD, GG>(p, max_extend_len, st);
#endif
#define M355_CASE(DD, GG) if (D == DD && G == GG) { \
        M355_PREFILL_LAUNCH(DD, GG) \
        if (p.splits > 1) { \
            const int64_t items = (int64_t)T * Hq; \
            if (f16) hipLaunchKernelGGL((prefill_merge_kernel<DD, true>), dim3((unsigned)((items * (DD / 4) + 255) / 256)), dim3(256), 0, st, \
                                        p.ws, p.out, p.o_stride, items, Hq, p.splits); \
            else hipLaunchKernelGGL((prefill_merge_kernel<DD>), dim3((unsigned)((items * (DD / 4) + 255) / 256)), dim3(256), 0, st, \
                                    p.ws, p.out, p.o_stride, items, Hq, p.splits); \
        } \
        return check_launch(); }
    M355_CASE(64, 1) M355_CASE(64, 2) M355_CASE(64, 4) M355_CASE(64, 8)
    M355_CASE(128, 1) M355_CASE(128, 2) M355_CASE(128, 4) M355_CASE(128, 8)
#undef M355_CASE
    return M355_ERR_UNSUPPORTED;
}
