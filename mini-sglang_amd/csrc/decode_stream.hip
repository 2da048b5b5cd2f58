// Planned ("stream-split") paged decode attention for gfx950.
//
// m355_decode_attn (decode_attn.hip) gives every (request, kv head, chunk) its own wave.  Measured on MI355X its
// loop phase streams K/V at ~6 TB/s, but each wave pays ~4.5 us of dependent loads before its first tile and
// ~3 us after its last, ragged batches leave dead blocks and a tail, and with one round of waves nothing hides
// any of it.  Here the work is laid out once per forward instead (the page tables are the same for all layers):
//
//   plan  (m355_decode_plan, 2 small launches per forward)
//     cum[b]       exclusive prefix sum of seq_lens: the batch's tokens form one flat stream of `total` tokens
//     R            tokens per range = ceil(total / waves-per-head); range r = flat tokens [r*R, (r+1)*R)
//     rec[r]       range record: length, and its PIECES (request, start inside the range, length, partial index) -
//                  a range crosses request boundaries
//     slots[r][i]  KV-pool slot of the i-th token of range r (page-table lookup done once, not once per layer)
//   attention (m355_decode_attn_planned, per layer): grid = Hkv x NWH single-wave workgroups, exactly filling the
//     chip; wave (h, w) walks ranges w, w+NWH, ...  Record and slots sit at addresses that depend on r only, so
//     they are fetched together by the wave's first loads; the K/V tile pipeline then runs across piece boundaries
//     without draining; q rows of all pieces are staged in LDS; a piece that covers its whole request writes bf16
//     output directly, the others write fp32 (m, l, acc) partials for the merge kernel.
//
// Every wave gets the same number of tokens, so there is one round, no tail and no dead blocks.
// Algorithmic bytes are those of decode_attn.hip; the plan adds 4 bytes per token per forward.
#include <stdlib.h>

#include <mutex>

#include "share_plan.h"

namespace m355 {

// Scratch of the allocation-free entry point (m355_decode_store_attn_planned_static): the fp32 partials workspace as a
// file-scope device array.  Every GPU the library runs on holds its own copy at its own address until the process
// ends; host code never uses the array's name as an address - only the per-device pointer that
// hipGetSymbolAddress returned (static_scratch() below).
static __device__ __align__(256) unsigned char scratch[M355_STATIC_SCRATCH_BYTES];

#ifdef M355_DEV
static unsigned long long* g_stream_stamps = nullptr;   // dev build: host copy of the timing probes' buffer pointer
#define M355_STREAM_STAMPS g_stream_stamps
#else
#define M355_STREAM_STAMPS nullptr
#endif

// ------------------------------------------------------------------------------------------------
// plan kernels
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int upper_bound_minus1(const int32_t* cum, int n, int32_t x) {
    // largest b in [0, n) with cum[b] <= x    (cum[0] == 0, non-decreasing)
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (cum[mid] <= x) lo = mid; else hi = mid - 1;
    }
    return lo;
}

__device__ __forceinline__ int32_t pieces_of(int32_t a, int32_t e, int32_t R) { return e > a ? (e - 1) / R - a / R + 1 : 0; }

constexpr int kCumLds = 4096;                       // requests whose cumulative lengths the plan kernels stage in LDS

__global__ __launch_bounds__(1024) void plan_scan_kernel(const int64_t* __restrict__ seq_lens, int32_t B, int32_t nwh, int32_t Hkv,
                                                         int32_t r_min, int64_t cap, int64_t wmax, int32_t* __restrict__ plan,
                                                         PlanLayout L) {
    __shared__ int32_t wsum[16];
    __shared__ int32_t cum_s[kCumLds + 1];                // cum of the first kCumLds requests: the record loop's binary
    __shared__ int32_t pb_s[kCumLds + 1];                 // searches and piece walks then stay out of global memory
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int32_t* cum = plan + L.cum;
    int32_t* pbase = plan + L.pbase;
    const int per = (B + 1023) / 1024;
    const int lo = tid * per, hi = min(B, lo + per);
    auto block_scan = [&](int32_t mine) -> int32_t {       // exclusive scan of one value per thread: 6 shuffles + 16 wave sums
        int32_t incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int32_t v = __shfl_up(incl, d);
            if (lane >= d) incl += v;
        }
        if (lane == 63) wsum[wv] = incl;
        __syncthreads();
        int32_t base = 0;
        for (int k = 0; k < wv; ++k) base += wsum[k];
        __syncthreads();
        return base + incl - mine;
    };
    const bool staged = B <= kCumLds;
    // a request longer than its page-table row is wide (wmax tokens) is cut at the row's end: nothing past the row is looked up
    auto len_of = [&](int b) -> int32_t { const int64_t l = seq_lens[b]; return (int32_t)(l < wmax ? (l > 0 ? l : 0) : wmax); };
    int32_t s = 0;
    for (int b = lo; b < hi; ++b) s += len_of(b);
    int32_t run = block_scan(s);
    // lengths are clamped so the stream never exceeds the slot slabs (a too-small capacity truncates, never overruns)
    const int32_t icap = (int32_t)cap;
    for (int b = lo; b < hi; ++b) {
        const int32_t c = run < icap ? run : icap;
        cum[b] = c;
        if (staged) cum_s[b] = c;
        run += len_of(b);
    }
    if (tid == 1023) { const int32_t c = run < icap ? run : icap; cum[B] = c; if (staged) cum_s[B] = c; }   // the grand total
    __syncthreads();
    const int32_t* cu = staged ? cum_s : cum;
    const int32_t total = cu[B];
    int32_t* full = plan + L.full;
    for (int b = lo; b < hi; ++b) full[b] = (int64_t)(cu[b + 1] - cu[b]) == seq_lens[b];
    if (tid == 1023) full[B] = 0;
    // one range per resident wave; when that would exceed kRMax tokens, k equal ranges per wave (k rounds of the
    // grid-stride loop) rather than kRMax-token ranges whose count is no multiple of the wave count
    const int32_t per_wave = (total + nwh - 1) / nwh;
    const int32_t rounds = (per_wave + kRMax - 1) / kRMax;
    int32_t R = rounds > 0 ? (per_wave + rounds - 1) / rounds : r_min;
    R = R < r_min ? r_min : R;
    const int32_t n_ranges = (total + R - 1) / R;
    if (tid == 0) { plan[0] = total; plan[1] = R; plan[2] = B; plan[3] = nwh; plan[4] = n_ranges; plan[5] = plan[6] = plan[7] = 0; }
    // partials per request -> pbase
    s = 0;
    for (int b = lo; b < hi; ++b) s += pieces_of(cu[b], cu[b + 1], R);
    run = block_scan(s);
    for (int b = lo; b < hi; ++b) {
        pbase[b] = run;
        if (staged) pb_s[b] = run;
        run += pieces_of(cu[b], cu[b + 1], R);
    }
    if (tid == 1023) { pbase[B] = run; if (staged) pb_s[B] = run; }
#ifdef M355_DEV      // (the arrival counts belong to the dev build's in-kernel merge; the product never looks at them)
    for (int64_t k = tid; k < ((int64_t)L.max_ranges + B) * Hkv; k += 1024) plan[L.arrivals + k] = 0;
#endif
    __syncthreads();
    const int32_t* pb = staged ? pb_s : pbase;
    // range records
    for (int r = tid; r < L.max_ranges; r += 1024) {
        int32_t* rec = plan + L.rec + (int64_t)r * kRec;
        if (r >= n_ranges) { rec[0] = 0; rec[1] = 0; rec[2] = 0; rec[3] = 0; continue; }
        const int32_t g0 = r * R, g1 = min(total, g0 + R);
        int b = upper_bound_minus1(cu, B, g0);
        rec[0] = g1 - g0; rec[2] = b; rec[3] = g0;
        int np = 0;
        int32_t endmask = 0;                                 // bit k: piece k ends its request (holds the newest token)
        for (; b < B && cu[b] < g1; ++b) {
            const int32_t cs = cu[b], ce = cu[b + 1];
            const int32_t lo2 = max(cs, g0), hi2 = min(ce, g1);
            if (hi2 <= lo2) continue;                        // empty request
            if (np == kMaxPieces) { np = -1; break; }        // too many: the attention kernel scans cum itself
            rec[4 + np * 4 + 0] = b;
            rec[4 + np * 4 + 1] = lo2 - g0;
            rec[4 + np * 4 + 2] = hi2 - lo2;
            const bool whole = lo2 == cs && hi2 == ce;
            rec[4 + np * 4 + 3] = whole ? -1 : pb[b] + (r - cs / R);
            rec[kRecCut + 2 * np] = whole ? 0 : pieces_of(cs, ce, R);
            rec[kRecCut + 2 * np + 1] = whole ? 0 : r - cs / R;
            if (hi2 == ce && (int64_t)(ce - cs) == seq_lens[b]) endmask |= 1 << np;   // (a cut request has no newest slot)
            ++np;
        }
        rec[1] = np;
        rec[kRecEnds] = endmask;
    }
}

__global__ __launch_bounds__(256) void plan_slots_kernel(const int32_t* __restrict__ page_table, int64_t pt_stride,
                                                         const int64_t* __restrict__ rows, int32_t B, int32_t ps,
                                                         int32_t* __restrict__ plan, PlanLayout L) {
    __shared__ int32_t cum_s[kCumLds + 1];
    const int32_t total = plan[0], R = plan[1];
    const int32_t* cum = plan + L.cum;
    if (B <= kCumLds) {                                      // the per-token binary search then runs in LDS
        for (int i = threadIdx.x; i <= B; i += blockDim.x) cum_s[i] = cum[i];
        __syncthreads();
        cum = cum_s;
    }
    int32_t* slabs = plan + L.slots;
    for (int32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < total; g += gridDim.x * blockDim.x) {
        const int b = upper_bound_minus1(cum, B, g);
        const int32_t t = g - cum[b];
        const int32_t pg = t / ps;
        const int32_t r = g / R;
        slabs[(int64_t)r * kRMax + (g - r * R)] = page_table[rows[b] * pt_stride + pg] * ps + (t - pg * ps);
    }
}

#ifdef M355_DEV      // ---- the VALU attention kernel over the plan: an A/B reference (M355_STREAM_KERNEL=valu), dev build only
// ------------------------------------------------------------------------------------------------
// attention
// ------------------------------------------------------------------------------------------------
template <int D, int G>
struct PieceState {
    uint32_t q[G][4];
    float m[G], l[G];
    float o[G][8];
};

template <int D, int G, int U>
__device__ __forceinline__ void stream_compute_tile(PieceState<D, G>& st, const u32x4 (&kt)[U], const u32x4 (&vt)[U],
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

template <int D, int G, int MASK>
__device__ __forceinline__ void merge_groups_step(PieceState<D, G>& st) {
    constexpr int LPH = D / 8;
    if constexpr (MASK >= LPH) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float ma, mb, la, lb;
            lane_pair<MASK>(st.m[g], ma, mb);
            lane_pair<MASK>(st.l[g], la, lb);
            const float mx = fmaxf(ma, mb);
            const float wa = exp2_fast(ma - mx), wb = exp2_fast(mb - mx);
            st.l[g] = la * wa + lb * wb;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float oa, ob;
                lane_pair<MASK>(st.o[g][j], oa, ob);
                st.o[g][j] = oa * wa + ob * wb;
            }
            st.m[g] = mx;
        }
    }
}

template <int D, int G, int U, int NB, int W, bool STAMP = false>
__global__ __launch_bounds__(64, W) void decode_stream_kernel(const StreamParams p) {
    constexpr int LPH = D / 8;
    constexpr int NT = kWave / LPH;
    constexpr int TPI = NT * U;
    constexpr int QCH = G * D / 8;                       // 16-byte chunks of one piece's q rows
    __shared__ __attribute__((aligned(16))) uint32_t lds[kRMax + 64 + kMaxPieces * QCH * 4];
    int32_t* slots = reinterpret_cast<int32_t*>(lds);
    int32_t* rec = slots + kRMax;                        // record copy; rec[4 + 4k ..] = piece k {b, start, len, pidx}
    int32_t* pc = rec + 4;
    u32x4* qs = reinterpret_cast<u32x4*>(lds + kRMax + 64);

    sstamp<STAMP>(p, 0);
    const int lane = threadIdx.x;
    const int dchunk = lane % LPH;
    const int tsub = lane / LPH;
    const int w = blockIdx.x / p.Hkv;
    const int h = (blockIdx.x + (p.rot > 0 ? w / p.rot : 0)) % p.Hkv;   // heads rotate over the XCDs every p.rot ranges (see decode_mfma.hip)
    const int32_t* plan = p.plan;
    const int64_t lane_off = (int64_t)h * D + dchunk * 8;

    PieceState<D, G> st;

    // One pass = the pieces currently in pc[0 .. np): stage q, then run the K/V tile pipeline across them.
    auto run_pass = [&](const int np) {
        int lk = 0, lit = 0;                              // load iterator: piece, tile
        int ck = 0, cit = 0;                              // compute iterator
        int l_start = __builtin_amdgcn_readfirstlane(pc[1]);
        int l_len = __builtin_amdgcn_readfirstlane(pc[2]);
        int c_len = l_len;

        auto load_tile = [&](u32x4 (&kt)[U], u32x4 (&vt)[U]) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                int tt = lit * TPI + u * NT + tsub;
                tt = tt < l_len ? tt : l_len - 1;         // clamp: the row is masked through the score bias
                const int64_t off = (int64_t)slots[l_start + tt] * p.row_elems + lane_off;
                kt[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p.k_pool + off));   // read once per step: see glds16p
                vt[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p.v_pool + off));
            }
            ++lit;
            if (lit * TPI >= l_len) {                     // next piece
                ++lk; lit = 0;
                if (lk < np) {
                    l_start = __builtin_amdgcn_readfirstlane(pc[lk * 4 + 1]);
                    l_len = __builtin_amdgcn_readfirstlane(pc[lk * 4 + 2]);
                }
            }
        };
        auto begin_piece = [&]() {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const u32x4 t = qs[ck * QCH + g * LPH + dchunk];
                st.q[g][0] = t[0]; st.q[g][1] = t[1]; st.q[g][2] = t[2]; st.q[g][3] = t[3];
                st.m[g] = kNegBig; st.l[g] = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) st.o[g][j] = 0.f;
            }
        };
        auto end_piece = [&]() {
            merge_groups_step<D, G, 8>(st);               // lane groups sit at lane bits >= log2(LPH)
            merge_groups_step<D, G, 16>(st);
            merge_groups_step<D, G, 32>(st);
            if (lane < LPH) {
                const int b = pc[ck * 4 + 0];
                const int pidx = pc[ck * 4 + 3];
                if (pidx < 0) {
                    uint16_t* op = p.out + (int64_t)b * p.o_stride + (int64_t)h * G * D + dchunk * 8;
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        const float inv = 1.f / st.l[g];
                        u32x4 wv;
#pragma unroll
                        for (int x = 0; x < 4; ++x) wv[x] = pack_bf16(st.o[g][2 * x] * inv, st.o[g][2 * x + 1] * inv);
                        *reinterpret_cast<u32x4*>(op + g * D) = wv;
                    }
                } else {
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        const int64_t idx = (int64_t)pidx * p.Hq + h * G + g;
                        float* wo = p.ws_o + idx * D + dchunk * 8;
                        *reinterpret_cast<f32x4*>(wo) = f32x4{st.o[g][0], st.o[g][1], st.o[g][2], st.o[g][3]};
                        *reinterpret_cast<f32x4*>(wo + 4) = f32x4{st.o[g][4], st.o[g][5], st.o[g][6], st.o[g][7]};
                        if (dchunk == 0) { p.ws_ml[idx * 2] = st.m[g]; p.ws_ml[idx * 2 + 1] = st.l[g]; }
                    }
                }
            }
        };

        // first K/V tiles go out before the q rows: both are one dependent level behind the record
        sstamp<STAMP>(p, 2);
        u32x4 kb[NB][U], vb[NB][U];
#pragma unroll
        for (int j = 0; j < NB - 1; ++j)
            if (lk < np) load_tile(kb[j], vb[j]);
        for (int i = lane; i < np * QCH; i += kWave) {
            const int k = i / QCH, c = i - k * QCH;
            qs[i] = *reinterpret_cast<const u32x4*>(p.q + (int64_t)pc[k * 4] * p.q_stride + (int64_t)h * G * D + c * 8);
        }
        __syncthreads();
        sstamp<STAMP>(p, 3);
        while (ck < np) {
#pragma unroll
            for (int j = 0; j < NB; ++j) {               // j static: the NB register tiles rotate by name
                if (ck < np) {
                    if (lk < np) load_tile(kb[(j + NB - 1) % NB], vb[(j + NB - 1) % NB]);
                    if (cit == 0) begin_piece();
                    float bias[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) bias[u] = (cit * TPI + u * NT + tsub) < c_len ? 0.f : kNegBig;
                    stream_compute_tile<D, G, U>(st, kb[j], vb[j], bias, p.scale_log2);
                    ++cit;
                    if (cit * TPI >= c_len) {
                        end_piece();
                        ++ck; cit = 0;
                        if (ck < np) c_len = __builtin_amdgcn_readfirstlane(pc[ck * 4 + 2]);
                    }
                }
            }
        }
        __syncthreads();                                 // rec / qs / slots are rewritten next
    };

    for (int r = w; r < p.max_ranges; r += p.nwh) {
        // ---- first loads of the wave: the record and (speculatively) the first kSpec*64 slots, in parallel
        const int32_t* grec = plan + p.off_rec + (int64_t)r * kRec;
        const int32_t* gslots = plan + p.off_slots + (int64_t)r * kRMax;
        int32_t rv = 0, sv[kSpec];
        if (lane < kRec) rv = grec[lane];
#pragma unroll
        for (int k = 0; k < kSpec; ++k) sv[k] = gslots[lane + kWave * k];
        if (lane < kRec) rec[lane] = rv;
#pragma unroll
        for (int k = 0; k < kSpec; ++k) slots[lane + kWave * k] = sv[k];
        __syncthreads();
        const int32_t n = __builtin_amdgcn_readfirstlane(rec[0]);
        const int32_t np_rec = __builtin_amdgcn_readfirstlane(rec[1]);
        if (n == 0) break;                               // past the last range
        if (n > kSpec * kWave) {
            for (int i = kSpec * kWave + lane; i < n; i += kWave) slots[i] = gslots[i];
            __syncthreads();
        }
        sstamp<STAMP>(p, 1);
        if (np_rec >= 0) {
            run_pass(np_rec);
        } else {
            // ---- more than kMaxPieces requests in this range (many tiny requests): scan cum in passes
            const int32_t* cum = plan + p.off_cum;
            const int32_t* pbase = plan + p.off_pbase;
            const int32_t g0 = __builtin_amdgcn_readfirstlane(rec[3]);
            const int32_t R = plan[1];
            int b_next = __builtin_amdgcn_readfirstlane(rec[2]);
            int32_t done = 0;
            __syncthreads();
            while (done < n) {
                const int b = b_next + lane;
                int32_t cs = 0, ce = 0;
                if (lane < kMaxPieces && b < p.B) { cs = cum[b]; ce = cum[b + 1]; }
                const int32_t lo = max(cs, g0 + done), hi = min(ce, g0 + n);
                const bool have = lane < kMaxPieces && b < p.B && hi > lo;
                const unsigned long long mask = __ballot(have);
                const int np = __popcll(mask);
                if (have) {
                    const int k = __popcll(mask & ((1ull << lane) - 1));
                    pc[k * 4 + 0] = b;
                    pc[k * 4 + 1] = lo - g0;
                    pc[k * 4 + 2] = hi - lo;
                    pc[k * 4 + 3] = (lo == cs && hi == ce) ? -1 : pbase[b] + (r - cs / R);
                }
                const int last = 63 - __clzll(mask | 1ull);
                done = __builtin_amdgcn_readfirstlane((np > 0 ? __shfl(hi, last) : g0 + n) - g0);
                b_next = __builtin_amdgcn_readfirstlane(np > 0 ? b_next + last + 1 : b_next + kMaxPieces);
                __syncthreads();
                if (np > 0) run_pass(np);
            }
        }
    }
    if constexpr (STAMP) { sstamp<STAMP>(p, 4); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); sstamp<STAMP>(p, 5); }
}

// One wave per (request, q head): combine the partials of requests cut by range boundaries.
// NW waves per (request, q head): wave v combines partials v, v + NW, ...; the NW results meet in LDS.  NW = 1 for
// batches whose requests are cut into a few pieces; long contexts in small batches (hundreds of pieces per request)
// would otherwise walk them in one serial chain of memory latencies.
#endif  // M355_DEV

template <int D, int NW, bool F16 = false>
__global__ __launch_bounds__(64 * NW) void stream_merge_kernel(const int32_t* plan_hot, const int64_t off_pbase_hot, const int64_t off_cum_hot,
                                                               const int32_t Hq_hot, const StreamParams p) {
    // (leading scalars: preloaded into SGPRs at wave launch, see decode_mfma.hip - the first loads leave at once)
    const int b = blockIdx.x / Hq_hot;
    const int hq = blockIdx.x % Hq_hot;
    const int32_t* pbase = plan_hot + off_pbase_hot;
    const int32_t* cum = plan_hot + off_cum_hot;
    const int32_t R = plan_hot[1];
    const int32_t cs = cum[b], ce = cum[b + 1];
    const int np = pieces_of(cs, ce, R);
    if (np <= 1) return;                                  // whole request handled by one wave: output already final
    const int32_t p0 = pbase[b];
    constexpr int E = D / kWave;
    const int lane = threadIdx.x & 63;
    const int wv = NW > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
    // All loads of a batch of <= 8 partials go out together (m, l and this lane's slice of o): one memory latency
    // per batch instead of three dependent ones.
    float mx = kNegBig, den = 0.f, acc[E];
#pragma unroll
    for (int e = 0; e < E; ++e) acc[e] = 0.f;
    const int np_w = (np - wv + NW - 1) / NW;            // partials of this wave: wv, wv + NW, ...
    for (int s0 = 0; s0 < np_w; s0 += 8) {
        float mm[8], ll[8], oo[8][E];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int sj = wv + NW * (s0 + j < np_w ? s0 + j : np_w - 1);
            const int64_t idx = (int64_t)(p0 + sj) * p.Hq + hq;
            mm[j] = p.ws_ml[idx * 2];
            ll[j] = p.ws_ml[idx * 2 + 1];
            const float* wo = p.ws_o + idx * D + lane * E;
#pragma unroll
            for (int e = 0; e < E; ++e) oo[j][e] = wo[e];
        }
        float bm = mx;
#pragma unroll
        for (int j = 0; j < 8; ++j) bm = fmaxf(bm, s0 + j < np_w ? mm[j] : kNegBig);
        const float resc = exp2_fast(mx - bm);
        den *= resc;
#pragma unroll
        for (int e = 0; e < E; ++e) acc[e] *= resc;
        mx = bm;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float wgt = s0 + j < np_w ? exp2_fast(mm[j] - mx) : 0.f;
            den = fmaf(wgt, ll[j], den);
#pragma unroll
            for (int e = 0; e < E; ++e) acc[e] = fmaf(wgt, oo[j][e], acc[e]);
        }
    }
    if constexpr (NW > 1) {                              // (mx, den, acc) of the NW waves -> wave 0
        __shared__ float red[NW][kWave * (E + 2)];
        red[wv][lane] = mx;                              // mx and den are wave-uniform; every lane writes its copy
        red[wv][kWave + lane] = den;
#pragma unroll
        for (int e = 0; e < E; ++e) red[wv][kWave * (2 + e) + lane] = acc[e];
        __syncthreads();
        if (wv != 0) return;
        float gm = kNegBig;
#pragma unroll
        for (int v = 0; v < NW; ++v) gm = fmaxf(gm, red[v][lane]);
        den = 0.f;
#pragma unroll
        for (int e = 0; e < E; ++e) acc[e] = 0.f;
#pragma unroll
        for (int v = 0; v < NW; ++v) {
            const float wgt = exp2_fast(red[v][lane] - gm);       // a wave with no partials has mx = -big, den = 0: weight 0
            den = fmaf(wgt, red[v][kWave + lane], den);
#pragma unroll
            for (int e = 0; e < E; ++e) acc[e] = fmaf(wgt, red[v][kWave * (2 + e) + lane], acc[e]);
        }
    }
    const float inv = 1.f / den;
    uint16_t* op = p.out + (int64_t)b * p.o_stride + (int64_t)hq * D + lane * E;
    if constexpr (E == 2) {
        *reinterpret_cast<uint32_t*>(op) = pack16<F16>(acc[0] * inv, acc[1] * inv);
    } else {
        *op = (uint16_t)(pack16<F16>(acc[0] * inv, 0.f) & 0xffffu);
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
#ifdef M355_DEV
// Launch shape of the VALU kernel per group size: {U, NB, waves per SIMD}.  M355_STREAM_VARIANT is a tuning hook (0 = default).
static int stream_variant() {
    static const int v = dev_env_int("M355_STREAM_VARIANT", 0);
    return v;
}
struct StreamShape { int U, NB, W; };
static inline StreamShape stream_shape(int G) {
    const int v = stream_variant();
    if (G >= 8) return v == 1 ? StreamShape{1, 4, 2} : StreamShape{2, 2, 2};
    if (G == 4) {
        switch (v) {
            case 1: return {2, 2, 3};
            case 2: return {2, 4, 2};
            case 3: return {2, 3, 3};
            default: return {2, 3, 2};
        }
    }
    return v == 1 ? StreamShape{2, 4, 3} : StreamShape{2, 3, 3};
}
// Which attention kernel consumes the plan: the matrix-core kernel (decode_mfma.hip), or in the dev build the VALU
// kernel above (M355_STREAM_KERNEL=valu).  The plan's waves-per-head and shortest range follow the kernel.
static bool use_mfma() {
    static const bool v = dev_env_char("M355_STREAM_KERNEL") != 'v';
    return v;
}
static inline int stream_nwh(int Hkv, int D, int G, int64_t cap) {
    const int wpc = use_mfma() ? mfma_waves_per_cu(Hkv, D, G, cap) : 4 * stream_shape(G).W;
    const int n = cu_count() * wpc / Hkv;
    return n < 1 ? 1 : n;
}
static inline int stream_rmin(int D, int G) {                                                       // >= 4 tiles per range
    return use_mfma() ? mfma_rmin() : 4 * (kWave / (D / 8)) * stream_shape(G).U;
}
#else
constexpr bool use_mfma() { return true; }
static inline int stream_nwh(int Hkv, int D, int G, int64_t cap) {
    const int n = cu_count() * mfma_waves_per_cu(Hkv, D, G, cap) / Hkv;
    return n < 1 ? 1 : n;
}
static inline int stream_rmin(int, int) { return mfma_rmin(); }
#endif

// Which kernel serves a batch: the single-launch share kernel (decode_share.hip) when, on average, a request spans at
// most kShareSpan shares (its finishing wave combines the parts of a cut request one after the other) AND a share is
// at least kShareMinBytes of K/V: below that the launch is a few microseconds of streaming and the publish -> fetch
// chains of cut requests outlast it (measured, dev build, both kernels forced: (8,1,128) B=64 at 148 KB per share 20.8 us
// against 17.2 for ranges + merge; ties from 300 to 600 KB; the share kernel ahead by 3-8 % from 600 KB up).  The
// per-wave ranges + merge launch (decode_mfma.hip) serves the rest: small launches, and few long requests (thousands of
// tokens per request and CU), where hundreds of pieces per request want the wide merge.  Plan, workspace and attention
// calls all decide from the same (B, head shape, token_capacity); the dev build can force either with M355_DECODE_SHARE=0/1.
// A third condition looks at the POOL: in a pool whose token rows are a multiple of 1 KiB apart (the reference's contiguous
// [S, Hkv, D] rows, memory/kvcache.py:24-35, for every power-of-two head shape) one kv head's bytes of ALL tokens share
// address bits 7..9, and a CU that streams only that 128-byte column (head_dim 64) runs well below its rate: the share
// kernel, one kv head per workgroup, takes 34.5 us on configs[1] there against 29.3 on rows padded by 128 bytes, while the
// range kernel - eight waves of a CU on eight different heads - takes 31.8 on either (profiles/r03_contiguous_ab.txt; at
// head_dim 128, 256-byte columns, the two are within 3 % of each other, either way).  Such pools go to the range kernel.
constexpr int kShareSpan = 8;
constexpr int64_t kShareMinBytes = 512 * 1024;
static inline int64_t pool_pitch(int32_t Hkv, int32_t D, int64_t pool_row_stride) { return pool_row_stride > 0 ? pool_row_stride : (int64_t)Hkv * D; }
static bool share_mode(int32_t B, int32_t Hkv, int32_t D, int32_t G, int64_t token_capacity, int64_t pool_row_stride) {
    if (!use_mfma()) return false;
    const ShareShape sh = share_shape(Hkv, D, G);
    if (!sh.ok) return false;
    static const int forced = dev_env_int("M355_DECODE_SHARE", -1);
    if (forced >= 0) return forced != 0;
    const bool pinned_columns = D == 64 && Hkv > 1 && (pool_pitch(Hkv, D, pool_row_stride) * 2) % 1024 == 0;
    return (int64_t)B * kShareSpan >= sh.nb && token_capacity * 4 * D >= (int64_t)sh.nb * kShareMinBytes && !pinned_columns;
}

}  // namespace m355

using namespace m355;

#ifdef M355_DEV
extern "C" int m355dbg_set_stream_stamp_buffer(void* dev_ptr) {   // dev build only (not part of include/mi355attn.h)
    m355::g_stream_stamps = (unsigned long long*)dev_ptr;
    return 0;
}
#endif

extern "C" int64_t m355_decode_plan_bytes(int32_t B, int32_t Hkv, int32_t D, int32_t G, int64_t token_capacity, int64_t pool_row_stride) {
    if (B <= 0 || Hkv <= 0 || token_capacity <= 0) return 0;
    if (share_mode(B, Hkv, D, G, token_capacity, pool_row_stride))
        return share_layout(B, share_shape(Hkv, D, G).nb, Hkv, token_capacity, D, G).ints * (int64_t)sizeof(int32_t);
    return plan_layout(B, stream_nwh(Hkv, D, G, token_capacity), Hkv, token_capacity).ints * (int64_t)sizeof(int32_t);
}

extern "C" int m355_decode_plan(const int64_t* seq_lens, const int64_t* page_table_ids, const int32_t* page_table,
                                int64_t pt_row_stride, int32_t pt_cols, int32_t B, int32_t page_size, int32_t Hkv, int32_t D, int32_t G,
                                int64_t token_capacity, int64_t pool_row_stride, void* plan, int64_t plan_bytes, void* stream) {
    if (B == 0) return M355_OK;
    if (!seq_lens || !page_table_ids || !page_table || !plan || B < 0 || page_size <= 0 || Hkv <= 0 || token_capacity <= 0 ||
        pt_row_stride <= 0 || pt_cols <= 0 || pt_cols > pt_row_stride)
        return M355_ERR_INVALID_ARG;
    if ((D != 64 && D != 128) || (G != 1 && G != 2 && G != 4 && G != 8)) return M355_ERR_UNSUPPORTED;
    if (token_capacity > 0x7fffffffLL / 2 || pool_row_stride < 0) return M355_ERR_INVALID_ARG;
    if (share_mode(B, Hkv, D, G, token_capacity, pool_row_stride)) {
        const ShareLayout SL = share_layout(B, share_shape(Hkv, D, G).nb, Hkv, token_capacity, D, G);
        if (plan_bytes < SL.ints * (int64_t)sizeof(int32_t) || ((uintptr_t)plan & 15)) return M355_ERR_WORKSPACE;
        return launch_share_plan(seq_lens, page_table_ids, page_table, pt_row_stride, pt_cols, B, page_size, Hkv, D, G, token_capacity,
                                 (int32_t*)plan, (hipStream_t)stream);
    }
    const int nwh = stream_nwh(Hkv, D, G, token_capacity);
    const PlanLayout L = plan_layout(B, nwh, Hkv, token_capacity);
    if (plan_bytes < L.ints * (int64_t)sizeof(int32_t) || ((uintptr_t)plan & 15)) return M355_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(plan_scan_kernel, dim3(1), dim3(1024), 0, st, seq_lens, B, nwh, Hkv, stream_rmin(D, G), token_capacity,
                       (int64_t)pt_cols * page_size, (int32_t*)plan, L);
    int64_t blocks = (token_capacity + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(plan_slots_kernel, dim3((unsigned)blocks), dim3(256), 0, st, page_table, pt_row_stride,
                       page_table_ids, B, page_size, (int32_t*)plan, L);
    return check_launch();
}

extern "C" int64_t m355_decode_planned_workspace_bytes(int32_t B, int32_t Hq, int32_t Hkv, int32_t D,
                                                       int64_t token_capacity, int64_t pool_row_stride) {
    if (B <= 0 || Hq <= 0 || Hkv <= 0 || D <= 0 || token_capacity <= 0) return 0;
    if (share_mode(B, Hkv, D, Hq / Hkv, token_capacity, pool_row_stride)) return 0;      // single launch: partials never leave LDS (hand-offs ride in the plan)
    const PlanLayout L = plan_layout(B, stream_nwh(Hkv, D, Hq / Hkv, token_capacity), Hkv, token_capacity);
    const int64_t n_partials = (int64_t)L.max_ranges + B;      // every range boundary and request adds at most one
    return n_partials * Hq * (D + 2) * (int64_t)sizeof(float);
}

// waves per (request, q head) of the merge: about one per 8 expected pieces of a request (ranges per head / requests)
template <int D, bool F16 = false>
static void launch_merge(const StreamParams& p, hipStream_t st) {
    const int per_req = p.nwh / (p.B > 0 ? p.B : 1);
    const dim3 grid((unsigned)(p.B * p.Hq));
    if (per_req >= 64) hipLaunchKernelGGL((stream_merge_kernel<D, 16, F16>), grid, dim3(1024), 0, st, p.plan, p.off_pbase, p.off_cum, p.Hq, p);
    else if (per_req >= 16) hipLaunchKernelGGL((stream_merge_kernel<D, 4, F16>), grid, dim3(256), 0, st, p.plan, p.off_pbase, p.off_cum, p.Hq, p);
    else hipLaunchKernelGGL((stream_merge_kernel<D, 1, F16>), grid, dim3(64), 0, st, p.plan, p.off_pbase, p.off_cum, p.Hq, p);
}

#ifdef M355_DEV
template <int D, int G>
static void launch_stream(const StreamParams& p, hipStream_t st) {
    const dim3 grid((unsigned)(p.Hkv * p.nwh)), block(64);
    const StreamShape sh = stream_shape(G);
    const bool stamp = dev_env_set("M355_STREAM_STAMP");
#define M355_SHAPE(UU, NN, WW) if (sh.U == UU && sh.NB == NN && sh.W == WW) { \
        if (stamp) hipLaunchKernelGGL((decode_stream_kernel<D, G, UU, NN, WW, true>), grid, block, 0, st, p); \
        else hipLaunchKernelGGL((decode_stream_kernel<D, G, UU, NN, WW>), grid, block, 0, st, p); }
    if constexpr (G >= 8) { M355_SHAPE(1, 4, 2) M355_SHAPE(2, 2, 2) }
    else if constexpr (G == 4) { M355_SHAPE(2, 2, 3) M355_SHAPE(2, 4, 2) M355_SHAPE(2, 3, 3) M355_SHAPE(2, 3, 2) }
    else { M355_SHAPE(2, 4, 3) M355_SHAPE(2, 3, 3) }
#undef M355_SHAPE
    launch_merge<D>(p, st);
}
#endif  // M355_DEV

static int decode_attn_planned_impl(const void* q, int64_t q_row_stride, const void* k_pool, const void* v_pool,
                                    int64_t pool_row_stride, void* plan, void* out, int64_t o_row_stride, int32_t B, int32_t Hq,
                                    int32_t Hkv, int32_t D, float scale, int64_t token_capacity, void* workspace,
                                    int64_t workspace_bytes, int32_t dtype, void* stream, const void* k_new,
                                    int64_t k_row_stride, const void* v_new, int64_t v_row_stride,
                                    bool library_scratch = false) {
    if (B == 0) return M355_OK;
    if (!q || !k_pool || !v_pool || !plan || !out || B < 0 || Hq <= 0 || Hkv <= 0 || Hq % Hkv || token_capacity <= 0)
        return M355_ERR_INVALID_ARG;
    const bool f16 = dtype == M355_DTYPE_F16;          // IEEE half: the matrix-core kernel only
    if (dtype != M355_DTYPE_BF16 && !(f16 && use_mfma())) return M355_ERR_UNSUPPORTED;
    if (D != 64 && D != 128) return M355_ERR_UNSUPPORTED;
    const int G = Hq / Hkv;
    if (G != 1 && G != 2 && G != 4 && G != 8) return M355_ERR_UNSUPPORTED;
    if (((uintptr_t)q | (uintptr_t)k_pool | (uintptr_t)v_pool | (uintptr_t)out | (uintptr_t)plan) & 15)
        return M355_ERR_INVALID_ARG;
    if ((q_row_stride % 8) || (o_row_stride % 8) || q_row_stride < (int64_t)Hq * D || o_row_stride < (int64_t)Hq * D)
        return M355_ERR_INVALID_ARG;
    if (pool_row_stride == 0) pool_row_stride = (int64_t)Hkv * D;
    if (pool_row_stride < (int64_t)Hkv * D || pool_row_stride % 8) return M355_ERR_INVALID_ARG;
    if (share_mode(B, Hkv, D, G, token_capacity, pool_row_stride)) {
        const ShareShape sh = share_shape(Hkv, D, G);
        const ShareLayout SL = share_layout(B, sh.nb, Hkv, token_capacity, D, G);
        ShareParams sp;
        sp.q = (const uint16_t*)q; sp.k_pool = (uint16_t*)k_pool; sp.v_pool = (uint16_t*)v_pool;
        sp.k_new = (const uint16_t*)k_new; sp.v_new = (const uint16_t*)v_new; sp.kn_stride = k_row_stride; sp.vn_stride = v_row_stride;
        sp.plan = (int32_t*)plan; sp.out = (uint16_t*)out;
        sp.q_stride = q_row_stride; sp.o_stride = o_row_stride; sp.row_elems = pool_row_stride;
        sp.off_stg = SL.stg; sp.off_share = SL.share; sp.off_gran = SL.gran; sp.off_rec = SL.rec; sp.qcap = SL.qcap;
        sp.B = B; sp.Hq = Hq; sp.Hkv = Hkv; sp.scale_log2 = scale * kLog2e;
        (void)library_scratch;
        sp.stamps = M355_STREAM_STAMPS;
        const int rc = launch_share_attn(sp, D, G, f16, (hipStream_t)stream, dev_env_set("M355_STREAM_STAMP"));
        return rc != M355_OK ? rc : check_launch();
    }
    const int nwh = stream_nwh(Hkv, D, G, token_capacity);
    const PlanLayout L = plan_layout(B, nwh, Hkv, token_capacity);
    const int64_t n_partials = (int64_t)L.max_ranges + B;
    const int64_t need = n_partials * Hq * (D + 2) * (int64_t)sizeof(float);
    if (!workspace || workspace_bytes < need || ((uintptr_t)workspace & 15)) return M355_ERR_WORKSPACE;
    StreamParams p;
    p.q = (const uint16_t*)q; p.k_pool = (const uint16_t*)k_pool; p.v_pool = (const uint16_t*)v_pool;
    p.plan = (const int32_t*)plan; p.out = (uint16_t*)out;
    p.k_new = (const uint16_t*)k_new; p.v_new = (const uint16_t*)v_new; p.kn_stride = k_row_stride; p.vn_stride = v_row_stride;
    p.ws_o = (float*)workspace; p.ws_ml = p.ws_o + n_partials * Hq * D;
    p.q_stride = q_row_stride; p.o_stride = o_row_stride; p.row_elems = pool_row_stride;
    p.off_cum = L.cum; p.off_pbase = L.pbase; p.off_full = L.full; p.off_arrivals = L.arrivals; p.off_rec = L.rec; p.off_slots = L.slots;
    // (the merging wave addresses the partials with 32-bit offsets)
    if (need >= (int64_t)1 << 31) return M355_ERR_UNSUPPORTED;
    { static const int in_kernel = dev_env_int("M355_MERGE_IN_KERNEL", 0);   // (dev build, 1: last-arriver merge inside the attention launch, A/B)
      p.merge_in_kernel = use_mfma() && in_kernel; }
    p.B = B; p.Hq = Hq; p.Hkv = Hkv; p.nwh = nwh; p.max_ranges = L.max_ranges; p.scale_log2 = scale * kLog2e;
    p.stamps = M355_STREAM_STAMPS;
    { static const int rot = dev_env_int("M355_STREAM_ROT", 1);    // (dev build, 0: no head rotation, for A/B)
      p.rot = rot ? (nwh / Hkv > 0 ? nwh / Hkv : 1) : 0; }
    hipStream_t st = (hipStream_t)stream;
    if (k_new && !use_mfma()) return M355_ERR_UNSUPPORTED;   // the fused store lives in the MFMA kernel only
    if (use_mfma()) {
        const int rc = launch_stream_mfma(p, D, G, token_capacity, st, dev_env_set("M355_STREAM_STAMP"), f16);
        if (rc != M355_OK) return rc;
        if (!p.merge_in_kernel) {
            if (f16) { if (D == 64) launch_merge<64, true>(p, st); else launch_merge<128, true>(p, st); }
            else { if (D == 64) launch_merge<64>(p, st); else launch_merge<128>(p, st); }
        }
        return check_launch();
    }
#ifdef M355_DEV
#define M355_CASE(DD, GG) if (D == DD && G == GG) { launch_stream<DD, GG>(p, st); return check_launch(); }
    M355_CASE(64, 1) M355_CASE(64, 2) M355_CASE(64, 4) M355_CASE(64, 8)
    M355_CASE(128, 1) M355_CASE(128, 2) M355_CASE(128, 4) M355_CASE(128, 8)
#undef M355_CASE
#endif
    return M355_ERR_UNSUPPORTED;
}

extern "C" int m355_decode_attn_planned(const void* q, int64_t q_row_stride, const void* k_pool, const void* v_pool,
                                        int64_t pool_row_stride, void* plan, void* out, int64_t o_row_stride, int32_t B, int32_t Hq,
                                        int32_t Hkv, int32_t D, float scale, int64_t token_capacity, void* workspace,
                                        int64_t workspace_bytes, int32_t dtype, void* stream) {
    return decode_attn_planned_impl(q, q_row_stride, k_pool, v_pool, pool_row_stride, plan, out, o_row_stride, B, Hq, Hkv, D, scale,
                                    token_capacity, workspace, workspace_bytes, dtype, stream, nullptr, 0, nullptr, 0);
}

// The device address of `scratch` on the CURRENT device, looked up once per device with that device current.
static void* static_scratch() {
    constexpr int kMaxDevices = 64;
    static std::mutex mu;
    static void* ptrs[kMaxDevices] = {};
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (!ptrs[dev]) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(m355::scratch)) != hipSuccess || !p) { (void)hipGetLastError(); return nullptr; }
        ptrs[dev] = p;
    }
    return ptrs[dev];
}

extern "C" int m355_decode_store_attn_planned_static(const void* q, int64_t q_row_stride, const void* k_new,
                                                     int64_t k_row_stride, const void* v_new, int64_t v_row_stride,
                                                     void* k_pool, void* v_pool, int64_t pool_row_stride, void* plan, void* out,
                                                     int64_t o_row_stride, int32_t B, int32_t Hq, int32_t Hkv, int32_t D,
                                                     float scale, int64_t token_capacity, int32_t dtype, void* stream) {
    if (B == 0) return M355_OK;
    if (!k_new || !v_new || k_row_stride < (int64_t)Hkv * D || v_row_stride < (int64_t)Hkv * D || (k_row_stride % 8) ||
        (v_row_stride % 8) || (((uintptr_t)k_new | (uintptr_t)v_new) & 15))
        return M355_ERR_INVALID_ARG;
    if (B < 0 || Hq <= 0 || Hkv <= 0 || D <= 0 || token_capacity <= 0) return M355_ERR_INVALID_ARG;
    if (m355_decode_planned_workspace_bytes(B, Hq, Hkv, D, token_capacity, pool_row_stride) > (int64_t)M355_STATIC_SCRATCH_BYTES)
        return M355_ERR_SCRATCH_LIMIT;
    void* ws = static_scratch();
    if (!ws) return M355_ERR_LAUNCH;
    // Nothing in the scratch has to start at zero: every partial the merge reads is written by this same call.
    return decode_attn_planned_impl(q, q_row_stride, k_pool, v_pool, pool_row_stride, plan, out, o_row_stride, B, Hq, Hkv, D, scale,
                                    token_capacity, ws, (int64_t)M355_STATIC_SCRATCH_BYTES, dtype, stream, k_new,
                                    k_row_stride, v_new, v_row_stride, /*library_scratch=*/true);
}

extern "C" int m355_decode_store_attn_planned(const void* q, int64_t q_row_stride, const void* k_new, int64_t k_row_stride,
                                              const void* v_new, int64_t v_row_stride, void* k_pool, void* v_pool,
                                              int64_t pool_row_stride, void* plan, void* out, int64_t o_row_stride, int32_t B, int32_t Hq,
                                              int32_t Hkv, int32_t D, float scale, int64_t token_capacity,
                                              void* workspace, int64_t workspace_bytes, int32_t dtype, void* stream) {
    if (B == 0) return M355_OK;
    if (!k_new || !v_new || k_row_stride < (int64_t)Hkv * D || v_row_stride < (int64_t)Hkv * D || (k_row_stride % 8) ||
        (v_row_stride % 8) || (((uintptr_t)k_new | (uintptr_t)v_new) & 15))
        return M355_ERR_INVALID_ARG;
    return decode_attn_planned_impl(q, q_row_stride, k_pool, v_pool, pool_row_stride, plan, out, o_row_stride, B, Hq, Hkv, D, scale,
                                    token_capacity, workspace, workspace_bytes, dtype, stream, k_new, k_row_stride, v_new,
                                    v_row_stride);
}

// ------------------------------------------------------------------------------------------------
// The self-contained form, for callers without a per-forward hook: lay the plan out and attend in ONE call.  The
// workspace holds the plan (for B * max_seq_len_hint tokens, the most the batch can hold) followed by the partials.
// ------------------------------------------------------------------------------------------------
static inline int64_t up256(int64_t x) { return (x + 255) & ~(int64_t)255; }

extern "C" int64_t m355_decode_workspace_bytes(int32_t B, int32_t Hq, int32_t Hkv, int32_t D, int32_t max_seq_len_hint,
                                               int64_t pool_row_stride) {
    if (B <= 0 || Hq <= 0 || Hkv <= 0 || D <= 0 || max_seq_len_hint <= 0 || Hq % Hkv) return 0;
    const int64_t cap = (int64_t)B * max_seq_len_hint;
    return up256(m355_decode_plan_bytes(B, Hkv, D, Hq / Hkv, cap, pool_row_stride)) +
           up256(m355_decode_planned_workspace_bytes(B, Hq, Hkv, D, cap, pool_row_stride));
}

extern "C" int m355_decode_attn(const void* q, int64_t q_row_stride, const void* k_pool, const void* v_pool,
                                int64_t pool_row_stride, const int32_t* page_table, int64_t pt_row_stride,
                                const int64_t* page_table_ids, const int64_t* seq_lens, void* out, int64_t o_row_stride,
                                int32_t B, int32_t Hq, int32_t Hkv, int32_t D, int32_t page_size, float scale,
                                int32_t max_seq_len_hint, void* workspace, int64_t workspace_bytes, int32_t dtype, void* stream) {
    if (B == 0) return M355_OK;
    if (B < 0 || Hq <= 0 || Hkv <= 0 || Hq % Hkv || max_seq_len_hint <= 0) return M355_ERR_INVALID_ARG;
    const int64_t cap = (int64_t)B * max_seq_len_hint;
    if (cap > 0x7fffffffLL / 2) return M355_ERR_INVALID_ARG;
    const int G = Hq / Hkv;
    if ((D != 64 && D != 128) || (G != 1 && G != 2 && G != 4 && G != 8)) return M355_ERR_UNSUPPORTED;
    if (pool_row_stride < 0) return M355_ERR_INVALID_ARG;
    const int64_t pb = up256(m355_decode_plan_bytes(B, Hkv, D, G, cap, pool_row_stride));
    const int64_t wb = up256(m355_decode_planned_workspace_bytes(B, Hq, Hkv, D, cap, pool_row_stride));
    if (!workspace || workspace_bytes < pb + wb || ((uintptr_t)workspace & 255)) return M355_ERR_WORKSPACE;
    // (this form is not told the table's width: its row stride bounds the look-ups, so they never leave the request's row)
    if (pt_row_stride <= 0 || pt_row_stride > 0x7fffffffLL) return M355_ERR_INVALID_ARG;
    const int rc = m355_decode_plan(seq_lens, page_table_ids, page_table, pt_row_stride, (int32_t)pt_row_stride, B, page_size, Hkv, D, G, cap,
                                    pool_row_stride, workspace, pb, stream);
    if (rc != M355_OK) return rc;
    return m355_decode_attn_planned(q, q_row_stride, k_pool, v_pool, pool_row_stride, workspace, out, o_row_stride, B, Hq, Hkv, D, scale,
                                    cap, (char*)workspace + pb, wb, dtype, stream);
}
