// Ragged prefill ("extend") attention, 64 query rows per wave at one wave per SIMD (reference
// torch_native_backend.py:25-119, 195-235; same arithmetic, tile images and plan as prefill_attn.hip).
//
// prefill_wg_kernel keeps two workgroups per CU so that one wave's softmax runs beside its partner's MFMAs: every K/V
// fragment is read from LDS, every tile brought in by LDS-DMA, every barrier paid once per 32 query rows, and the SIMD's
// vector issue port, shared by the two waves, is what fills up.  Here a wave owns TWO 32-row tiles and the
// whole 512-register file: a K or V^T fragment feeds two MFMAs; DMA pieces, LDS reads, slot look-ups and barriers halve
// per MFMA; and the overlap comes from the wave's own instruction stream - the softmax of one row tile is issued in the
// gaps between the MFMAs of the other.
//
// hipcc does not keep such an order by itself (it sinks the exponentials behind the MFMA block they should run between).
// The order is therefore PINNED: the stream is cut into 64 windows per pair of KV tiles, one MFMA each, by empty
// `asm volatile` statements that take the MFMA's fragment ("+a": it lives in the accumulator half of the register file)
// and the running values of the softmax chunk ("+v") as read-write operands; volatile statements keep their order, and
// the data dependences through them hold every MFMA and every chunk inside its window.  No instruction is written in
// assembly: register allocation, wait counts and hazard padding stay the compiler's.
//
//   windows  1- 8  S0a = Ka Q0^T        next pair's K pieces; K fragments of tile b
//            9-16  S0b = Kb Q0^T        max of S0a
//           17-24  S1a = Ka Q1^T        max of S0b, decision 0, exp2 of S0a (first part)
//           25-32  S1b = Kb Q1^T        exp2 of S0a / S0b; V^T fragments of tile a
//           33-40  O0 += Va^T P0a^T     exp2 of S0b (rest), max of S1, decision 1; V^T fragments of tile b
//           41-48  O0 += Vb^T P0b^T     exp2 of S1a
//           49-56  O1 += Va^T P1a^T     exp2 of S1b
//           57-64  O1 += Vb^T P1b^T     next pair: wait + barrier + K fragments of its tile a, V pieces
#include "prefill_common.h"

#ifndef M355_WG2_EXPERIMENT
#define M355_WG2_EXPERIMENT 0     // timing ablations for tools/ (WRONG results): 1 no steady-state DMA, 2 no pair barrier, 3 no softmax chunks, 4 = 1+3, 5 = 4 + no fragment reads
#endif

namespace m355 {

template <int D, int G, int NBUF, bool POW2, bool F16>
__global__ __launch_bounds__(256, 1) void prefill_wg2_kernel(const PrefillParams p) {
    using W = WgCfg<D>;
    constexpr int NW = 4, RT = 2;
    constexpr int NT = 64 * NW;
    constexpr int IPW = 2 * W::NI / NW;
    constexpr int NP = NBUF / 2;
    static_assert(W::KS == 8 && W::DT == 4, "the window plan below is laid out for head_dim 128");
    static_assert(NBUF % 2 == 0 && NP >= 2 && NP <= 4 && (2 * W::NI) % NW == 0, "ring geometry");
    extern __shared__ __attribute__((aligned(1024))) uint8_t smem[];
    int32_t* const slotw = reinterpret_cast<int32_t*>(smem + NBUF * W::BUFB);
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int l = threadIdx.x & 63;
    const int c = l & 31, hh = l >> 5;
    constexpr int ROWS_W = 32 * RT, ROWS_WG = ROWS_W * NW;

    const int n_bh = p.B * p.Hkv;                                  // XCD-aware order: see prefill_wg_kernel
    const int mapped = (n_bh >> 3) * 8 * p.tiles_max;
    int tb, bh;
    if ((int)blockIdx.x < mapped) {
        const int item = blockIdx.x >> 3;
        tb = p.tiles_max - 1 - item % p.tiles_max;
        bh = (item / p.tiles_max) * 8 + (blockIdx.x & 7);
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
    if (ROWS_WG * tb >= rows_b) return;
    const int32_t* pt_row = p.page_table + p.page_table_ids[b] * p.pt_stride;
    const int64_t qrow0 = p.cu_q[b];
    const int j_last_wg = min(E - 1, (ROWS_WG * tb + ROWS_WG - 1) / G);
    const int kv_end_wg = P + j_last_wg + 1;
    const int n_tiles = (kv_end_wg + 31) >> 5;
    const int n_pairs_all = (n_tiles + 1) >> 1;
    const int p_begin = p.splits > 1 ? (int)((int64_t)n_pairs_all * blockIdx.y / p.splits) : 0;
    const int n_pairs = p.splits > 1 ? (int)((int64_t)n_pairs_all * (blockIdx.y + 1) / p.splits) : n_pairs_all;

    // ---- this wave's two row tiles
    const int t = tb * NW + wid;                                   // 64-row wave tile inside the request
    bool valid[RT];
    int jrow[RT], grow[RT], limit[RT];
    bf16x8 qf[RT][W::KS];
    f32x16 o[RT][W::DT];
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        const int mrow = ROWS_W * t + 32 * r + c;
        valid[r] = mrow < rows_b;
        jrow[r] = mrow / G; grow[r] = mrow % G;
        limit[r] = valid[r] ? P + jrow[r] + 1 : 0;
        const uint16_t* qp = p.q + (qrow0 + jrow[r]) * p.q_stride + (int64_t)(h * G + grow[r]) * D + hh * 8;
#pragma unroll
        for (int s = 0; s < W::KS; ++s) {
            u32x4 v = u32x4{0u, 0u, 0u, 0u};
            if (valid[r]) v = *reinterpret_cast<const u32x4*>(qp + s * 16);
            qf[r][s] = __builtin_bit_cast(bf16x8, v);
        }
#pragma unroll
        for (int dt = 0; dt < W::DT; ++dt)
#pragma unroll
            for (int i = 0; i < 16; ++i) o[r][dt][i] = 0.f;
    }
    const int kv_end = ROWS_W * t < rows_b ? P + min(E - 1, (ROWS_W * t + ROWS_W - 1) / G) + 1 : 0;   // wave-uniform
    const int kv_min = P + (ROWS_W * t) / G + 1;
    constexpr float kRunFloor = -1.0e20f;                          // (see prefill_wg_kernel)
    float m_run[RT], l_run[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) { m_run[r] = kRunFloor; l_run[r] = 0.f; }

    // ---- DMA duty (as prefill_wg_kernel)
    const int drow = l / W::C, dch = l % W::C;
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)smem);
    const uint8_t* kbase = reinterpret_cast<const uint8_t*>(p.k_pool) + (int64_t)h * W::ROWB;
    const uint8_t* vbase = reinterpret_cast<const uint8_t*>(p.v_pool) + (int64_t)h * W::ROWB;
    const uint32_t row_bytes = (uint32_t)(p.row_elems * 2);
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
    static_assert(IPW % 2 == 0 && W::NI % NW == 0, "a wave's K pieces and V pieces cover the same rows");
    constexpr int HP = IPW / 2;
    auto pair_slots = [&](int pair, int32_t (&sl)[IPW]) {
        pair = pair < n_pairs ? pair : n_pairs - 1;
        pair = pair > p_begin ? pair : p_begin;
        pair = pair < n_pairs_all ? pair : n_pairs_all - 1;
        if ((2 * pair) / (kSlotWin / 32) != cur_win) fill_slots((2 * pair) / (kSlotWin / 32));   // workgroup-uniform
#pragma unroll
        for (int x = 0; x < IPW; ++x) {
            const int tile = 2 * pair + x / HP < n_tiles ? 2 * pair + x / HP : n_tiles - 1;
            sl[x] = slotw[((tile * 32) & (kSlotWin - 1)) + (wid + NW * (x % HP)) * W::RPI + drow];
        }
    };
    uint32_t kch[HP], vch[HP];
#pragma unroll
    for (int y = 0; y < HP; ++y) {
        const int row = (wid + NW * y) * W::RPI + drow;
        kch[y] = 16 * (dch ^ W::sK(row));
        vch[y] = 16 * (dch ^ W::sV(row));
    }
    bool steady = false;
    auto piece = [&](const bool is_v, const int x, const int32_t slot, const int buf0) {
        if ((M355_WG2_EXPERIMENT == 1 || M355_WG2_EXPERIMENT >= 4) && steady) return;
        const int i = wid + NW * (x % HP);
        const uint64_t off = (uint64_t)(uint32_t)slot * row_bytes + (is_v ? vch[x % HP] : kch[x % HP]);
        glds16p<M355_PREFILL_POLICY>((is_v ? vbase : kbase) + off,
               lds0 + (is_v ? NBUF * W::TILEB : 0) + (buf0 + x / HP) * W::TILEB + 1024 * i);
    };
    auto issue_pair = [&](const int pair, const int buf0) {
        int32_t sl[IPW];
        pair_slots(pair, sl);
#pragma unroll
        for (int x = 0; x < IPW; ++x) { piece(false, x, sl[x], buf0); piece(true, x, sl[x], buf0); }
    };

    uint32_t kaddr[W::KS], vaddr[W::DT];
    {
        const int ksw = W::sK(c);
#pragma unroll
        for (int s = 0; s < W::KS; ++s) kaddr[s] = lds0 + c * W::ROWB + 16 * ((2 * s + hh) ^ ksw);
        const int gidx = l >> 4, li = l & 15;
        const int vrow = 4 * hh + (li >> 2);
        const int vcol_chunk = 2 * (gidx & 1) + ((li & 3) >> 1);
#pragma unroll
        for (int dt = 0; dt < W::DT; ++dt)
            vaddr[dt] = lds0 + NBUF * W::TILEB + vrow * W::ROWB + 16 * ((4 * dt + vcol_chunk) ^ W::sV(vrow)) + 8 * (li & 1);
    }
    static_assert(2 * NBUF * W::TILEB + kSlotWin * 4 <= 160 * 1024, "ring fits the LDS");

    auto rdk = [&](const int buf, const int s) {
        if (M355_WG2_EXPERIMENT == 5 && steady) return qf[0][s];
        return *reinterpret_cast<const bf16x8 __attribute__((address_space(3)))*>((uintptr_t)(kaddr[s] + buf * W::TILEB));
    };
    auto rdv = [&](const int buf, const int f) {
        if (M355_WG2_EXPERIMENT == 5 && steady) return qf[1][f];
        const uint32_t a0 = vaddr[f >> 1] + buf * W::TILEB + 16 * (f & 1) * W::ROWB;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(uintptr_t)a0);
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(uintptr_t)(a0 + 8 * W::ROWB));
        return __builtin_bit_cast(bf16x8, s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
    };
    auto mask = [&](f32x16& s, const int base, const int lim) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int tok = base + (i & 3) + 8 * (i >> 2) + 4 * hh;
            s[i] = tok < lim ? s[i] : kNegBig;
        }
    };
    auto rescale = [&](const int r, const float alpha) {
        if (__any(alpha != 1.f)) {
#pragma unroll
            for (int dt = 0; dt < W::DT; ++dt)
#pragma unroll
                for (int i = 0; i < 16; ++i) o[r][dt][i] *= alpha;
        }
    };

    // ---- the pins.  A window = { open; one MFMA; one chunk of softmax; close }.  `frag` is the MFMA's A operand: read by the
    // MFMA, "rewritten" by the closing pin, so the MFMA stays between the two; the chunk's carried values do the same for
    // the chunk.  (Volatile asm statements keep their relative order.)
#define M355_PIN_A(frag) asm volatile("" : "+a"(frag))
#define M355_PIN_V1(a) asm volatile("" : "+v"(a))
#define M355_PIN_V2(a, b) asm volatile("" : "+v"(a), "+v"(b))
#define M355_PIN_V3(a, b, c) asm volatile("" : "+v"(a), "+v"(b), "+v"(c))

    // softmax chunks of one row tile (state in sm_*): max over half a score tile; the decision; two exponentials
    float sm_mx, sm_negm, sm_sum, sm_alpha;
    uint32_t sm_w[16];
    auto chunk_max = [&](const f32x16& s, const int half, const bool first) {      // 8 elements: 4 x v_max3
        float mx = first ? s[8 * half] : sm_mx;
#pragma unroll
        for (int i = first ? 1 : 0; i < 8; ++i) mx = fmaxf(mx, s[8 * half + i]);
        sm_mx = mx;
        M355_PIN_V1(sm_mx);
    };
    auto chunk_decide = [&](const int r) {
        float ma, mb;
        lane_pair<32>(sm_mx, ma, mb);
        const float m_tile = fmaxf(ma, mb) * p.scale_log2;
        const float m_new = m_tile > m_run[r] + kLazy ? m_tile : m_run[r];
        sm_alpha = exp2_fast(m_run[r] - m_new);
        m_run[r] = m_new;
        sm_negm = -m_new;
        sm_sum = 0.f;
        M355_PIN_V3(sm_negm, sm_sum, sm_alpha);
    };
    auto chunk_exp = [&](f32x16& s, const int j, const int wbase) {               // elements 2j, 2j+1 -> one packed word
        if (M355_WG2_EXPERIMENT >= 3) { sm_w[wbase + j] = __builtin_bit_cast(uint32_t, s[2 * j]); M355_PIN_V3(sm_negm, sm_sum, sm_w[wbase + j]); return; }
        const float e0 = exp2_fast(fmaf(s[2 * j], p.scale_log2, sm_negm));
        const float e1 = exp2_fast(fmaf(s[2 * j + 1], p.scale_log2, sm_negm));
        sm_sum += e0 + e1;
        sm_w[wbase + j] = pack16<F16>(e0, e1);
        M355_PIN_V3(sm_negm, sm_sum, sm_w[wbase + j]);
    };

    // ---- prologue: NP-1 pairs in flight, the first one landed, its K fragments of tile a in registers
    wait_vmcnt<0>();
#pragma unroll
    for (int j = 0; j < NP - 1; ++j) issue_pair(p_begin + j, 2 * j);
    bf16x8 fa[W::KS], fb[W::KS];                                   // X and Y: K fragments, then V^T fragments (header comment)
    wait_vmcnt<(NP - 2) * 2 * IPW>();
    __syncthreads();
#pragma unroll
    for (int s = 0; s < W::KS; ++s) fa[s] = rdk(0, s);

    steady = true;
    auto pair_step = [&](const int k, auto pb_const) {
        constexpr int pb = decltype(pb_const)::value;
        constexpr int nb = 2 * ((pb + NP - 1) % NP);               // buffers that pair k-1 vacated
        constexpr int pn = (pb + 1) % NP;                          // ring slot of pair k+1
        int32_t sl[IPW];
        pair_slots(k + NP - 1, sl);
        const int base_a = 64 * k;
        if (base_a >= kv_end) {                                    // wave-uniform: both row tiles are past their diagonal
#pragma unroll
            for (int x = 0; x < IPW; ++x) { piece(false, x, sl[x], nb); piece(true, x, sl[x], nb); }
            wait_vmcnt<(NP - 2) * 2 * IPW>();
            __syncthreads();
            return;
        }
        const bool diag = base_a + 64 > kv_min;
        bf16x8 p0a[2], p0b[2], p1a[2], p1b[2];
        f32x16 s0a, s0b, s1a, s1b;
#pragma unroll
        for (int i = 0; i < 16; ++i) { s0a[i] = 0.f; s0b[i] = 0.f; s1a[i] = 0.f; s1b[i] = 0.f; }
#pragma unroll
        for (int s = 0; s < W::KS; ++s) fb[s] = rdk(2 * pb + 1, s);
        // ---- windows 1-8: S0a; the next pair's K pieces
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            M355_PIN_A(fa[s]);
            s0a = mfma_32x32x16<F16>(fa[s], qf[0][s], s0a);
            if (s >= 1 && s - 1 < IPW) piece(false, s - 1, sl[s - 1], nb);
            M355_PIN_A(fa[s]);
        }
        if (diag) mask(s0a, base_a, limit[0]);                     // wave-uniform: diagonal pairs only
        // ---- windows 9-16: S0b; max of S0a
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            M355_PIN_A(fb[s]);
            s0b = mfma_32x32x16<F16>(fb[s], qf[0][s], s0b);
            if (s == 4) chunk_max(s0a, 0, true);
            if (s == 5) chunk_max(s0a, 1, false);
            M355_PIN_A(fb[s]);
        }
        if (diag) mask(s0b, base_a + 32, limit[0]);
        // ---- windows 17-24: S1a; max of S0b, decision 0, exp2 of S0a (elements 0-9)
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            M355_PIN_A(fa[s]);
            s1a = mfma_32x32x16<F16>(fa[s], qf[1][s], s1a);
            if (s == 0) chunk_max(s0b, 0, false);
            if (s == 1) chunk_max(s0b, 1, false);
            if (s == 2) chunk_decide(0);
            if (s >= 3) chunk_exp(s0a, s - 3, 0);                   // j = 0..4
            M355_PIN_A(fa[s]);
        }
        // ---- windows 25-32: S1b; exp2 of S0a (rest) and S0b (first part); X takes the V^T fragments of tile a
        M355_PIN_V1(vaddr[0]);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            M355_PIN_A(fb[s]);
            s1b = mfma_32x32x16<F16>(fb[s], qf[1][s], s1b);
            fa[s] = rdv(2 * pb, s);
            if (s < 3) chunk_exp(s0a, 5 + s, 0);                    // j = 5..7
            else chunk_exp(s0b, s - 3, 8);                          // j = 0..4
            M355_PIN_A(fb[s]);
        }
        if (diag) { mask(s1a, base_a, limit[1]); mask(s1b, base_a + 32, limit[1]); }
        // ---- windows 33-40: O0 += Va P0a; exp2 of S0b (rest), max of S1, decision 1; Y takes the V^T fragments of tile b
        {
            // P0a is complete (exp2 of S0a ended in window 27); rescale O0 before its first PV product
            const float alpha0 = sm_alpha;
            rescale(0, alpha0);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            p0a[ks] = __builtin_bit_cast(bf16x8, u32x4{sm_w[4 * ks], sm_w[4 * ks + 1], sm_w[4 * ks + 2], sm_w[4 * ks + 3]});
        float l0_sum = 0.f, l0_alpha = 1.f;
        M355_PIN_V1(vaddr[1]);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            M355_PIN_A(fa[s]);
            o[0][s >> 1] = mfma_32x32x16<F16>(fa[s], p0a[s & 1], o[0][s >> 1]);
            fb[s] = rdv(2 * pb + 1, s);
            if (s < 3) chunk_exp(s0b, 5 + s, 8);                    // j = 5..7: softmax 0 complete
            if (s == 3) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    p0b[ks] = __builtin_bit_cast(bf16x8, u32x4{sm_w[8 + 4 * ks], sm_w[8 + 4 * ks + 1], sm_w[8 + 4 * ks + 2], sm_w[8 + 4 * ks + 3]});
                l0_sum = sm_sum; l0_alpha = sm_alpha;
                chunk_max(s1a, 0, true);
            }
            if (s == 4) chunk_max(s1a, 1, false);
            if (s == 5) chunk_max(s1b, 0, false);
            if (s == 6) chunk_max(s1b, 1, false);
            if (s == 7) chunk_decide(1);
            M355_PIN_A(fa[s]);
        }
        l_run[0] = fmaf(l_run[0], l0_alpha, l0_sum);
        // ---- windows 41-48: O0 += Vb P0b; exp2 of S1a
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            M355_PIN_A(fb[s]);
            o[0][s >> 1] = mfma_32x32x16<F16>(fb[s], p0b[s & 1], o[0][s >> 1]);
            chunk_exp(s1a, s, 0);
            M355_PIN_A(fb[s]);
        }
        rescale(1, sm_alpha);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            p1a[ks] = __builtin_bit_cast(bf16x8, u32x4{sm_w[4 * ks], sm_w[4 * ks + 1], sm_w[4 * ks + 2], sm_w[4 * ks + 3]});
        // ---- windows 49-56: O1 += Va P1a; exp2 of S1b
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            M355_PIN_A(fa[s]);
            o[1][s >> 1] = mfma_32x32x16<F16>(fa[s], p1a[s & 1], o[1][s >> 1]);
            chunk_exp(s1b, s, 8);
            M355_PIN_A(fa[s]);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            p1b[ks] = __builtin_bit_cast(bf16x8, u32x4{sm_w[8 + 4 * ks], sm_w[8 + 4 * ks + 1], sm_w[8 + 4 * ks + 2], sm_w[8 + 4 * ks + 3]});
        l_run[1] = fmaf(l_run[1], sm_alpha, sm_sum);
        // ---- windows 57-64: O1 += Vb P1b; the next pair: V pieces, wait, barrier, K fragments of its tile a into X
#pragma unroll
        for (int x = 0; x < IPW; ++x) piece(true, x, sl[x], nb);
        if (M355_WG2_EXPERIMENT != 2) {
        wait_vmcnt<(NP - 2) * 2 * IPW>();                          // pair k+1 has landed (this wave's pieces) ...
        __syncthreads();                                           // ... everyone's; everyone has read pair k's tiles
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            M355_PIN_A(fb[s]);
            o[1][s >> 1] = mfma_32x32x16<F16>(fb[s], p1b[s & 1], o[1][s >> 1]);
            fa[s] = rdk(2 * pn, s);
            M355_PIN_A(fb[s]);
        }
    };
    for (int k0 = p_begin; k0 < n_pairs; k0 += NP) {
        pair_step(k0, IntC<0>{});
        if (k0 + 1 < n_pairs) pair_step(k0 + 1, IntC<1>{});
        if constexpr (NP > 2) { if (k0 + 2 < n_pairs) pair_step(k0 + 2, IntC<2>{}); }
        if constexpr (NP > 3) { if (k0 + 3 < n_pairs) pair_step(k0 + 3, IntC<3>{}); }
    }
    wait_vmcnt<0>();                                               // the clamped re-fetches must not outlive the wave

#pragma unroll
    for (int r = 0; r < RT; ++r) {
        float la, lb;
        lane_pair<32>(l_run[r], la, lb);
        if (p.splits > 1) {
            if (valid[r]) {
                float* wo = p.ws + (((qrow0 + jrow[r]) * p.Hq + (h * G + grow[r])) * p.splits + blockIdx.y) * (D + 4);
#pragma unroll
                for (int dt = 0; dt < W::DT; ++dt)
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4)
                        *reinterpret_cast<f32x4*>(wo + 32 * dt + 8 * q4 + 4 * hh) =
                            f32x4{o[r][dt][4 * q4 + 0], o[r][dt][4 * q4 + 1], o[r][dt][4 * q4 + 2], o[r][dt][4 * q4 + 3]};
                if (hh == 0) { wo[D] = m_run[r]; wo[D + 1] = la + lb; }
            }
            continue;
        }
        const float inv = 1.f / (la + lb);
        if (valid[r]) {
            uint16_t* op = p.out + (qrow0 + jrow[r]) * p.o_stride + (int64_t)(h * G + grow[r]) * D + 4 * hh;
#pragma unroll
            for (int dt = 0; dt < W::DT; ++dt)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    uint2 w;
                    w.x = pack16<F16>(o[r][dt][4 * q4 + 0] * inv, o[r][dt][4 * q4 + 1] * inv);
                    w.y = pack16<F16>(o[r][dt][4 * q4 + 2] * inv, o[r][dt][4 * q4 + 3] * inv);
                    *reinterpret_cast<uint2*>(op + 32 * dt + 8 * q4) = w;
                }
        }
    }
#undef M355_PIN_A
#undef M355_PIN_V1
#undef M355_PIN_V2
#undef M355_PIN_V3
}

template <int G>
static void launch_wg2(PrefillParams p, int max_extend_len, hipStream_t st) {
    constexpr int D = 128, NBUF = 8;
    p.tiles_max = (max_extend_len * G + 255) / 256;
    const dim3 grid((unsigned)((int64_t)p.B * p.Hkv * p.tiles_max), (unsigned)p.splits), block(256);
    const size_t lds = NBUF * WgCfg<D>::BUFB + kSlotWin * 4;
    static bool attr_done = [] {
        (void)hipFuncSetAttribute((const void*)prefill_wg2_kernel<D, G, NBUF, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)prefill_wg2_kernel<D, G, NBUF, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)prefill_wg2_kernel<D, G, NBUF, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)prefill_wg2_kernel<D, G, NBUF, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        return true; }();
    (void)attr_done;
    if (p.f16) {
        if (p.ps_shift >= 0) hipLaunchKernelGGL((prefill_wg2_kernel<D, G, NBUF, true, true>), grid, block, lds, st, p);
        else hipLaunchKernelGGL((prefill_wg2_kernel<D, G, NBUF, false, true>), grid, block, lds, st, p);
    } else {
        if (p.ps_shift >= 0) hipLaunchKernelGGL((prefill_wg2_kernel<D, G, NBUF, true, false>), grid, block, lds, st, p);
        else hipLaunchKernelGGL((prefill_wg2_kernel<D, G, NBUF, false, false>), grid, block, lds, st, p);
    }
}

bool launch_prefill_wg2(const PrefillParams& p, int D, int G, int max_extend_len, hipStream_t st) {
    if (D != 128) return false;
    switch (G) {
        case 1: launch_wg2<1>(p, max_extend_len, st); return true;
        case 2: launch_wg2<2>(p, max_extend_len, st); return true;
        case 4: launch_wg2<4>(p, max_extend_len, st); return true;
        case 8: launch_wg2<8>(p, max_extend_len, st); return true;
        default: return false;
    }
}

}  // namespace m355
