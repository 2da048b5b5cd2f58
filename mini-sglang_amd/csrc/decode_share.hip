// Single-launch paged decode attention for gfx950: one workgroup per CU owns a SHARE of one kv head's token stream.
//
// decode_mfma.hip gives every wave its own range of the stream and needs a second launch (stream_merge_kernel) to
// combine the fp32 partials of requests that a range boundary cut - 4.3 us of a 33 us pair on BASELINE configs[1].
// Here the unit of scheduling is a CU-resident workgroup:
//
//   plan (share_plan_kernel + share_fill_kernel, once per forward, shared by all layers and kv heads)
//     the batch's tokens form one flat stream, cut into nb = CUs / Hkv equal SHARES; the requests (or parts of
//     requests) inside a share are its SEGMENTS, walked LAST SEGMENT FIRST; every segment is cut into 32-token
//     STAGES and the share's stage list is dealt round-robin to the workgroup's W streaming waves.  A stage has a
//     40-word record: the 32 KV-pool slots (page table resolved once per forward), request, valid tokens, flags.
//   attention (decode_share_kernel, once per layer): W streaming waves + 1 finishing wave per workgroup
//     streaming wave   per stage: record (2 stages ahead), q rows and K/V tiles arrive by LDS-DMA - a uniform number
//                      of DMA instructions per stage, so ONE static vmcnt drives the whole loop, across requests;
//                      the math is decode_mfma.hip's (v_mfma_f32_16x16x32, in-register P, transposed V reads);
//                      at its last stage of a segment the wave leaves its fp32 (m, l, O) in LDS and counts itself in
//     finishing wave   per segment, in order: waits for the segment's waves, merges their partials from LDS and
//                      writes the bf16 output - no second launch, no partials in HBM.
//   a request cut by a SHARE boundary (about half of them on configs[1]): the share holding its head part walks that
//     segment FIRST and publishes one merged partial (write-through stores, then a flag); the share holding its tail
//     part walks that segment LAST, and its finishing wave - idle until then - has already fetched the neighbour's
//     partial when its own waves arrive.  Only requests longer than a share leave a wait at the end.
//     Waits only ever point at LOWER workgroup indices; polls are bounded (plan header word kHError).
//
// Algorithmic bytes are those of decode_mfma.hip; the plan costs 160 B per 32 tokens per forward.  No workspace.
#include <stdlib.h>

#include <atomic>

#include "share_plan.h"

namespace m355 {

__device__ __forceinline__ int ub_minus1(const int32_t* cum, int n, int32_t x) {
    // largest b in [0, n) with cum[b] <= x    (cum[0] == 0, non-decreasing): skips empty requests
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (cum[mid] <= x) lo = mid; else hi = mid - 1;
    }
    return lo;
}
__device__ __forceinline__ int32_t stages_of(int32_t tokens) { return (tokens + kStageTok - 1) / kStageTok; }
// shares 0 .. rem-1 hold Q + 1 stages of the stream, the others Q
__device__ __forceinline__ int32_t share_start(int32_t i, int32_t Q, int32_t rem) { return i * Q + (i < rem ? i : rem); }
__device__ __forceinline__ int32_t share_of(int32_t v, int32_t Q, int32_t rem) {
    const int32_t big = rem * (Q + 1);
    return v < big ? v / (Q + 1) : rem + (v - big) / Q;
}

// ------------------------------------------------------------------------------------------------
// plan
//
// Coordinates: request b has stages_of(len_b) stages; laid end to end they form the STAGE stream, stage v of which is
// 32 consecutive tokens of one request counted from that request's start (its last stage may be partial).  Shares are
// equal pieces of Q stages of that stream, so every workgroup streams the same number of K/V tiles, and a request is
// only ever cut at a multiple of 32 of its own tokens (no cut adds a partial stage).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void share_plan_kernel(const int64_t* __restrict__ seq_lens, int32_t B, int32_t nb,
                                                          int32_t Hkv, int32_t W, int64_t cap, int64_t wmax, int32_t* __restrict__ plan,
                                                          ShareLayout L) {
    __shared__ int32_t wsum[16];
    __shared__ int32_t cum_s[kShareCumLds + 1];
    __shared__ int32_t stg_s[kShareCumLds + 1];
    __shared__ int32_t nz_s[kShareCumLds + 1];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    auto block_scan = [&](int32_t mine) -> int32_t {       // exclusive scan of one value per thread
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
    const bool staged = B <= kShareCumLds;
    int32_t* cum = plan + L.cum;
    int32_t* stg = plan + L.stg;
    int32_t* nz = plan + L.nz;
    const int per = (B + 1023) / 1024;
    const int lo = tid * per, hi = min(B, lo + per);
    // ---- cum: exclusive prefix sums of the lengths, clamped so the stream never exceeds the capacity; a request longer
    // than its page-table row is wide (wmax tokens) is cut at the row's end: nothing past the row is ever looked up
    auto len_of = [&](int b) -> int32_t { const int64_t l = seq_lens[b]; return (int32_t)(l < wmax ? (l > 0 ? l : 0) : wmax); };
    int32_t s = 0;
    for (int b = lo; b < hi; ++b) s += len_of(b);
    int32_t run = block_scan(s);
    const int32_t icap = (int32_t)cap;
    for (int b = lo; b < hi; ++b) {
        const int32_t c = run < icap ? run : icap;
        cum[b] = c;
        if (staged) cum_s[b] = c;
        run += len_of(b);
    }
    if (tid == 1023) { const int32_t c = run < icap ? run : icap; cum[B] = c; if (staged) cum_s[B] = c; }
    __syncthreads();
    const int32_t* cu = staged ? cum_s : cum;
    // ---- stg / nz: exclusive prefix sums of stages per request and of "request is not empty"
    s = 0;
    int32_t z = 0;
    for (int b = lo; b < hi; ++b) { const int32_t len = cu[b + 1] - cu[b]; s += stages_of(len); z += len > 0; }
    int32_t run_s = block_scan(s);
    int32_t run_z = block_scan(z);
    for (int b = lo; b < hi; ++b) {
        const int32_t len = cu[b + 1] - cu[b];
        stg[b] = run_s; nz[b] = run_z;
        if (staged) { stg_s[b] = run_s; nz_s[b] = run_z; }
        run_s += stages_of(len); run_z += len > 0;
    }
    if (tid == 1023) { stg[B] = run_s; nz[B] = run_z; if (staged) { stg_s[B] = run_s; nz_s[B] = run_z; } }
    __syncthreads();
    const int32_t* sg = staged ? stg_s : stg;
    const int32_t TS = sg[B];
    // shares of Q or Q + 1 stages (the first `rem` get the extra one): every workgroup streams within one stage of the
    // same amount.  A stream too short for that is dealt in shares of kShareMinStages, the last workgroups idle.
    int32_t Q = TS / nb, rem = TS - Q * nb;
    if (Q < kShareMinStages) { Q = kShareMinStages; rem = 0; }
    // ---- shares (nb <= 1024: one per thread): the first and the last request that have stages in it
    if (tid < nb) {
        int32_t* e = plan + L.share + (int64_t)tid * kShareW;
        const int64_t s0 = share_start(tid, Q, rem);
        int32_t bf = 0, bl = -1, early = 0;
        if (s0 < TS) {
            const int32_t s1 = (int32_t)min((int64_t)TS, s0 + Q + (tid < rem));
            bf = ub_minus1(sg, B, (int32_t)s0);
            bl = ub_minus1(sg, B, s1 - 1);
            // A share's segments are walked last request first, so the tail of a request that began in an earlier share (it
            // waits for that share's hand-off) comes last.  When that tail is SHORT it ends in the same round of stages as the
            // segment before it and the finishing wave serves the two one after the other - 0.8 us more behind the last stage,
            // on exactly the workgroups that end the launch.  Such a tail is walked SECOND instead (behind the segment that is
            // published): it is done early, its partial waits in the finishing wave's registers for the hand-off.
            // (Also tried: tails of up to 16 stages, and walking the whole requests in between in the direction that ends with
            // the longer one - no further gain on configs[1] or the 8B shape.)
            const int32_t* nzp = staged ? nz_s : nz;
            if (bl > bf && bl - bf < 63 && sg[bf] < (int32_t)s0 && min(sg[bf + 1], s1) - (int32_t)s0 <= kShareEarlyStages &&
                nzp[bl] - nzp[bf] >= 2)
                early = 1;
        }
        e[0] = bf; e[1] = bl; e[2] = early;
        for (int k = 3; k < kShareW; ++k) e[k] = 0;
    }
    if (tid == 0) {
        plan[kHTotal] = cu[B]; plan[kHShareStages] = Q; plan[kHB] = B; plan[kHShares] = nb; plan[kHWaves] = W;
        plan[kHStages] = TS; plan[kHError] = 0; plan[kHShareRem] = rem;
        for (int k = kHShareRem + 1; k < kShareHdr; ++k) plan[k] = 0;
    }
}

__global__ __launch_bounds__(256) void share_fill_kernel(const int32_t* __restrict__ page_table, int64_t pt_stride,
                                                         const int64_t* __restrict__ rows, const int64_t* __restrict__ seq_lens,
                                                         int32_t B, int32_t ps, int32_t W, int32_t* __restrict__ plan, ShareLayout L) {
    __shared__ int32_t cum_s[kShareCumLds + 1];
    const int32_t T = plan[kHTotal], Q = plan[kHShareStages], TS = plan[kHStages], rem = plan[kHShareRem];
    const int32_t* cum = plan + L.cum;
    const int32_t* stg = plan + L.stg;
    const int32_t* nz = plan + L.nz;
    if (B <= kShareCumLds) {
        for (int k = threadIdx.x; k <= B; k += blockDim.x) cum_s[k] = cum[k];
        __syncthreads();
        cum = cum_s;
    }
    // hand-off granules start lowered (the attention kernel lowers what it consumed; this covers a fresh buffer)
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < L.gran_words; k += (int64_t)gridDim.x * blockDim.x) plan[L.gran + k] = 0;
    for (int32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < T; g += gridDim.x * blockDim.x) {
        const int b = ub_minus1(cum, B, g);
        const int32_t tok = g - cum[b], len = cum[b + 1] - cum[b];
        const int32_t sb0 = stg[b], sb1 = stg[b + 1];
        const int32_t v = sb0 + tok / kStageTok, idx = tok % kStageTok;      // stage of the stream, place inside it
        const int32_t i = share_of(v, Q, rem);
        const int32_t s0 = share_start(i, Q, rem), s1 = min(TS, s0 + Q + (i < rem));
        const int32_t a = max(sb0, s0), e = min(sb1, s1);                     // the segment = this request's stages in the share
        const int32_t st = v - a, seg_n = e - a;
        const int32_t* she = plan + L.share + (int64_t)i * kShareW;
        const int32_t bf_i = she[0], bl = she[1];
        int32_t sl = (s1 - e) + st;                                           // the share's segments are walked last to first
        int32_t ord = nz[bl] - nz[b];
        if (she[2]) {                                                         // ... except a short first one, which is walked second
            if (b == bf_i) { sl = (s1 - max(stg[bl], s0)) + st; ord = 1; }
            else if (b != bl) { sl += min(stg[bf_i + 1], s1) - s0; ord += 1; }
        }
        int32_t* rec = plan + L.rec + ((int64_t)i * L.qcap + sl) * kRecW;
        const int32_t pg = tok / ps;
        const int32_t slot = page_table[rows[b] * pt_stride + pg] * ps + (tok - pg * ps);
        rec[idx] = slot;
        if (idx == 0) {
            int32_t meta = min(kStageTok, len - (tok / kStageTok) * kStageTok);
            if (st < W) meta |= kRecFirst;                              // the owning wave's previous stage lies before the segment
            if (st + W >= seg_n) meta |= kRecLast;                      // ... its next one behind it
            // the request's last stage holds its newest token - unless the request was cut (capacity, table width): then the
            // fused store has no slot to write and leaves the pools alone
            if (v == sb1 - 1 && (int64_t)len == seq_lens[b]) meta |= kRecNewest;
            rec[kRecB] = b;
            rec[kRecMeta] = meta;
            rec[kRecOrd] = ord;
            for (int k = kRecOrd + 1; k < kRecW; ++k) rec[k] = 0;
        }
        if (tok == len - 1)                                             // pad the request's last stage with its last slot
            for (int k = idx + 1; k < kStageTok; ++k) rec[k] = slot;
    }
}

// ------------------------------------------------------------------------------------------------
// attention
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void glds4(const void* gsrc, uint32_t lds_dst) {     // 4 bytes per lane -> lds_dst + 4*lane
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ float ld_agent(const float* p) {          // global_load_dword sc1: served past this CU's L1
    return __builtin_bit_cast(float, __hip_atomic_load(reinterpret_cast<const uint32_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st_agent(float* p, float v) {        // global_store_dword sc1: write-through
    __hip_atomic_store(reinterpret_cast<uint32_t*>(p), __builtin_bit_cast(uint32_t, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent16(u32x4* p, u32x4 v) {      // global_store_dwordx4 sc1: ONE write-through store of a granule
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ int32_t lds_peek(const int32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

#ifndef M355_CROSS_POL
#define M355_CROSS_POL 5          // cache policy of the hand-off's LDS-DMA loads (common.h glds16p): sc1, agent scope
#endif
#ifndef M355_CROSS_EVERY
#define M355_CROSS_EVERY 6        // polls (~0.2 us each) between two requests for a publisher's granules
#endif
constexpr int kPollLimit = 1 << 22;             // bounded waits (each round sleeps): ~seconds, then kHError is raised

template <int D, int G>
struct SG {                                      // per-wave LDS of a streaming wave, and the partial format
    using M = MG<D>;
    static constexpr int QB = G * M::ROWB;                       // bytes of one request's q rows for this head group
    static constexpr int QBP = (QB + 63) / 64 * 64;
    static constexpr int QL = QB / 16;                           // 16-byte lanes of them
    static constexpr int NQ = (QL + 63) / 64;                    // DMA instructions for them
    static constexpr int RECBYTES = kRecW * 4;
    static constexpr int WAVEB = 2 * M::STAGEB + 4 * RECBYTES + 2 * QBP;
    static constexpr int PW = (G * (D + 2) + 3) / 4 * 4;         // floats of one wave's partial in LDS: O[G][D], m[G], l[G]
    static constexpr int VEC = G * D / 64 > 0 ? G * D / 64 : 1;  // output elements per lane of the finishing wave
};

constexpr int share_max_waves(int D) { return D <= 64 ? kShareMaxWaves : 4; }     // (LDS: a D = 128 wave needs 35 KiB)

template <bool ON>
__device__ __forceinline__ void shstamp(const ShareParams& p, int wave, int k) {
    if constexpr (ON)
        if (p.stamps && (threadIdx.x & 63) == 0)
            p.stamps[((size_t)blockIdx.x * (kShareMaxWaves + 1) + wave) * 8 + k] = __builtin_amdgcn_s_memrealtime();
}

// The leading scalar arguments repeat six fields of `p`: what a wave needs to ask for its first stage records.  Scalars in
// front of the argument block are PRELOADED into SGPRs at wave launch (Makefile: -amdgpu-kernarg-preload-count), so the
// record requests go out without first waiting for a load of the arguments themselves; `p` (a by-reference block that is
// not preloaded) arrives meanwhile.
template <int D, int G, bool F16, bool STAMP = false>
__global__ __launch_bounds__(64 * (share_max_waves(D) + 1)) void decode_share_kernel(const int32_t* plan_hot, const int64_t off_rec_hot,
                                                                                    const int32_t qcap_hot, const int32_t W_hot,
                                                                                    const int32_t Hkv_hot, const int32_t rot_hot,
                                                                                    const ShareParams p) {
    using M = MG<D>;
    using S = SG<D, G>;
    static_assert(G <= 16 && G * D >= 64, "group shape");
    extern __shared__ __attribute__((aligned(1024))) uint8_t smem[];
    const int W = W_hot;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    float* const parts = reinterpret_cast<float*>(smem + W * S::WAVEB);
    // ctl[1]: segments the finishing wave has merged (a partial slot is free again once its segment is);
    // ctl[2 + slot]: waves that have left their partial of the segment currently using that slot
    int32_t* const ctl = reinterpret_cast<int32_t*>(parts + p.nslot * W * S::PW);
    shstamp<STAMP>(p, wave, 0);
    if (threadIdx.x < 8) ctl[threadIdx.x] = 0;           // (the one workgroup barrier that publishes this comes below, behind each
                                                         // wave's first loads)
    const int share = blockIdx.x / Hkv_hot;
    // heads rotate over the XCDs (workgroups are dealt round-robin to them): see decode_mfma.hip
    const int h = (blockIdx.x + (rot_hot > 0 ? share / rot_hot : 0)) % Hkv_hot;
    // a streaming wave's first two stage records sit at places known from the kernel arguments alone: asked for at once,
    // they travel beside the header loads below instead of behind them
    const int32_t* const grec = plan_hot + off_rec_hot + ((int64_t)share * qcap_hot + (wave < W ? wave : 0)) * kRecW;
    const uint32_t rec0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)smem) + (wave < W ? wave : 0) * S::WAVEB + 2 * M::STAGEB;
    if (wave < W && lane < kRecW) {
        glds4(grec + lane, rec0);
        glds4(grec + (int64_t)W * kRecW + lane, rec0 + S::RECBYTES);
    }
    const int Q = __builtin_amdgcn_readfirstlane(plan_hot[kHShareStages]);
    const int TS = __builtin_amdgcn_readfirstlane(plan_hot[kHStages]);
    const int rem = __builtin_amdgcn_readfirstlane(plan_hot[kHShareRem]);
    const int SO = share_start(share, Q, rem);                       // the share = stages [SO, SO + NS) of the stream
    const int NS = TS - SO < Q + (share < rem) ? TS - SO : Q + (share < rem);
    if (NS <= 0) { wait_vmcnt<0>(); return; }                        // (no LDS-DMA may outlive its wave)
    shstamp<STAMP>(p, wave, 1);

    if (wave == W) {
        // ======================================================================================== finishing wave
        __syncthreads();
        const int32_t* sh = p.plan + p.off_share + (int64_t)share * kShareW;
        const int bf = __builtin_amdgcn_readfirstlane(sh[0]);
        const int bl = __builtin_amdgcn_readfirstlane(sh[1]);
        const int early = __builtin_amdgcn_readfirstlane(sh[2]);    // the share's first request (a short tail) is walked second
        const int32_t* stg = p.plan + p.off_stg;
        const int S1 = SO + NS;
        constexpr int VEC = S::VEC;
        constexpr int NG = share_granules(D, G);                     // 16-byte granules per lane: {3 of its VEC + 2 values (O, m, l), tag}
        // hand-off area of (kv head h, share i): NG x 64 granules, granule k of lane L at (k*64 + L)
        u32x4* const gran = reinterpret_cast<u32x4*>(p.plan + p.off_gran) + (int64_t)h * p.nb * (NG * 64);
        const int e0 = lane * VEC;                                   // first of this lane's elements of O[G][D]
        const int g = e0 / D, d0 = e0 - g * D;
        // hand-off state: only the share's FIRST request can end a request that began in an earlier share
        int cross_left = 0, cross_next = share - 1;
        {
            const int f0 = __builtin_amdgcn_readfirstlane(stg[bf]);
            if (f0 < SO && __builtin_amdgcn_readfirstlane(stg[bf + 1]) <= S1) cross_left = share - share_of(f0, Q, rem);   // the shares from f0's up to share-1 publish
        }
        float cm = kNegBig, cl = 0.f, cacc[VEC];
#pragma unroll
        for (int x = 0; x < VEC; ++x) cacc[x] = 0.f;
        bool failed = false;
        // The hand-off runs BESIDE the segment loop and never makes it wait.  A publisher's partial travels as 16-byte
        // granules {3 values, tag = publisher's share + 1}, each written by ONE write-through store, so there is no flag and
        // no ordering to wait for.  This wave asks for the next publisher's granules by LDS-DMA into a staging area of its
        // own every kCrossEvery polls and LOOKS at the staging area (LDS reads) on every poll: nothing here ever waits for
        // memory.  (Loaded into registers, the granules were loop-carried values: the register copies hipcc places on the
        // loop's edges each waited for the loads in flight - ~2 us per segment on a streaming CU, during which arrived
        // segments went unmerged and hand-offs unnoticed: 57 of 248 merging workgroups on configs[1] ended on it.)
        // All tags raised: the partial is folded into (cm, cl, cacc) and its granules are lowered again for the next
        // launch.  A DMA that lands after that carries the old publisher's tag and is ignored.
        constexpr int kCrossEvery = M355_CROSS_EVERY;
        uint8_t* const stage = reinterpret_cast<uint8_t*>(ctl) + 64;
        const uint32_t stage0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)stage);
#pragma unroll
        for (int k = 0; k < NG; ++k) *reinterpret_cast<u32x4*>(stage + k * 1024 + 16 * lane) = u32x4{0u, 0u, 0u, 0u};
        int cage = kCrossEvery;
        auto store_out = [&](const int b, const float den, const float (&acc)[VEC]) {
            const float inv = __builtin_amdgcn_rcpf(den);            // (one instruction, 1 ulp; the division is a dozen)
            uint16_t* op = p.out + (int64_t)b * p.o_stride + (int64_t)(h * G + g) * D + d0;
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
        };
        // a merge segment that was walked early and found the hand-off still open: its own (m, l, O) wait here
        bool pend_m = false;
        int pend_b = 0;
        float pend_gm = 0.f, pend_den = 0.f, pend_acc[VEC];
#pragma unroll
        for (int x = 0; x < VEC; ++x) pend_acc[x] = 0.f;
        auto finish_merge = [&](const int b, const float gm, float den, float (&acc)[VEC]) {   // hand-off (cm, cl, cacc) + own part -> output
            const float nm = fmaxf(cm, gm);
            const float wa = exp2_fast(cm - nm), wb = exp2_fast(gm - nm);
            den = cl * wa + den * wb;
#pragma unroll
            for (int x = 0; x < VEC; ++x) acc[x] = cacc[x] * wa + acc[x] * wb;
            store_out(b, den, acc);
        };
        auto cross_step = [&]() {
            const u32x4* src = gran + (int64_t)cross_next * (NG * 64) + lane;
            if (++cage >= kCrossEvery) {                             // ask (again): served past this CU's L1 (sc1)
                cage = 0;
#pragma unroll
                for (int k = 0; k < NG; ++k) glds16p<M355_CROSS_POL>(src + k * 64, stage0 + k * 1024);
            }
            const uint32_t want = (uint32_t)cross_next + 1u;
            bool mine = true;
#pragma unroll
            for (int k = 0; k < NG; ++k)
                mine = mine && __hip_atomic_load(reinterpret_cast<const uint32_t*>(stage + k * 1024 + 16 * lane + 12), __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_WORKGROUP) == want;
            if (!__all(mine)) return;                                // not (all) there yet: look again next time
            __builtin_amdgcn_s_sleep(1);                             // (a granule lands as one LDS write; this covers one that does not)
            asm volatile("" ::: "memory");
            float vals[3 * NG];
#pragma unroll
            for (int k = 0; k < NG; ++k)
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    vals[3 * k + j] = __builtin_bit_cast(float, __hip_atomic_load(reinterpret_cast<const uint32_t*>(stage + k * 1024 + 16 * lane + 4 * j),
                                                                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            shstamp<STAMP>(p, wave, 6);
            const float pm = vals[VEC], pl = vals[VEC + 1];
            const float nm = fmaxf(cm, pm);
            const float wa = exp2_fast(cm - nm), wb = exp2_fast(pm - nm);
            cl = cl * wa + pl * wb;
#pragma unroll
            for (int x = 0; x < VEC; ++x) cacc[x] = cacc[x] * wa + vals[x] * wb;
            cm = nm;
#pragma unroll
            for (int k = 0; k < NG; ++k) st_agent16(const_cast<u32x4*>(src) + k * 64, u32x4{0u, 0u, 0u, 0u});
            --cross_next; --cross_left;
            cage = kCrossEvery;                                      // the next publisher's granules are asked for at once
            if (cross_left == 0) {
                shstamp<STAMP>(p, wave, 3);
                if (pend_m) { finish_merge(pend_b, pend_gm, pend_den, pend_acc); pend_m = false; }
            }
        };
        // the segments, last request first; their stage ranges come from stg, 64 requests per load
        int base = 0, ord = 0;
        int wb = bl + 1;                                             // lanes hold requests wb-1, wb-2, ... of the current window
        int32_t w0 = 0, w1 = 0;
        // The window is loaded OUTSIDE the segment loop (and again, rarely, after 64 segments), and the loaded values are made
        // to be "used" right there: hipcc then waits for the loads there.  With the first load inside the loop it waited at the
        // top of EVERY iteration - s_waitcnt vmcnt(0), which also waits for the previous segment's output store (~0.8 us) and
        // for whatever hand-off DMA is in flight (2-3.5 us on a streaming chip): two segments that complete together were
        // served 0.8 us apart (the workgroups that ended a launch: shares whose last segment is a stage or two), and a
        // segment that arrived behind a hand-off request waited for the request.
        auto load_window = [&](const int b) {
            wb = b + 1;
            const int bb = b - lane;
            w0 = bb >= bf ? stg[bb] : 0;
            w1 = bb >= bf ? stg[bb + 1] : 0;
            asm volatile("" : "+v"(w0), "+v"(w1));
        };
        load_window(bl);
        const int nreq = bl - bf + 1;
        for (int k = 0; k < nreq; ++k) {
            // walk order: bl, bl-1, ..., bf - or, with a short first request, bl, bf, bl-1, ..., bf+1 (all inside one window then)
            const int b = early ? (k == 0 ? bl : (k == 1 ? bf : bl - (k - 1))) : bl - k;
            if (b < wb - 63) load_window(b);
            const int sb0 = __shfl(w0, wb - 1 - b), sb1 = __shfl(w1, wb - 1 - b);
            if (sb1 == sb0) continue;                                // empty request
            const int lo_ = sb0 > SO ? sb0 : SO, hi_ = sb1 < S1 ? sb1 : S1;
            const int nst = hi_ - lo_;
            const int kind = sb1 > S1 ? kSegPublish : (sb0 < SO ? kSegMerge : kSegWhole);
            const int e = nst < W ? nst : W;
            const int first = base % W;
            base += nst;
            int32_t* const arrived = ctl + 2 + ord % p.nslot;
            int spins = 0;
            while (lds_peek(arrived) < e) {                          // the segment's waves; meanwhile, the hand-off
                if (cross_left > 0) cross_step();
                __builtin_amdgcn_s_sleep(2);
                if (++spins > kPollLimit) { failed = true; break; }
            }
            asm volatile("" ::: "memory");
            if (ord == 0) shstamp<STAMP>(p, wave, 2);
            if (k == nreq - 1) shstamp<STAMP>(p, wave, 4);
            // All partials of the segment are requested at once, as whole vectors and from addresses that hold a partial for
            // EVERY w (a wave that took no part stands in wave `first`'s and gets weight 0): no branches, ONE LDS latency.
            // (Selecting the loaded values instead made hipcc branch around each load and wait for each wave's reads in
            // turn: 0.8 us per segment, and on the last segment of the last workgroup that is the launch's tail.)
            const float* slot = parts + (ord % p.nslot) * W * S::PW;
            constexpr int WMAX = share_max_waves(D);
            typedef float fvec __attribute__((ext_vector_type(VEC)));
            float pm[WMAX], pl[WMAX];
            fvec pa[WMAX];
#pragma unroll
            for (int w = 0; w < WMAX; ++w) {
                int rel = w - first;
                rel += rel < 0 ? W : 0;
                const bool part = w < W && rel < e;                  // wave w streamed a piece of this segment
                const float* pw = slot + (part ? w : first) * S::PW;
                pa[w] = *reinterpret_cast<const fvec*>(pw + e0);
                const float m_ = pw[G * D + g];
                pl[w] = pw[G * D + G + g];
                pm[w] = part ? m_ : kNegBig;
            }
            float gm = pm[0];
#pragma unroll
            for (int w = 1; w < WMAX; ++w) gm = fmaxf(gm, pm[w]);
            float den = 0.f, acc[VEC];
#pragma unroll
            for (int x = 0; x < VEC; ++x) acc[x] = 0.f;
#pragma unroll
            for (int w = 0; w < WMAX; ++w) {
                const float wgt = exp2_fast(pm[w] - gm);             // (absent waves: 2^(-1e30 - gm) = 0, times a finite stand-in)
                den = fmaf(wgt, pl[w], den);
#pragma unroll
                for (int x = 0; x < VEC; ++x) acc[x] = fmaf(wgt, pa[w][x], acc[x]);
            }
#ifdef M355_TAIL_STAMPS
            if (k == nreq - 1) shstamp<STAMP>(p, wave, 6);
#endif
            if (lane == 0) {                                         // slot free: count back to zero first, then release
                __hip_atomic_store(arrived, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __hip_atomic_store(ctl + 1, ord + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            if (kind == kSegPublish) {                               // every lane: its elements, and m / l of its q head
                u32x4* dst = gran + (int64_t)share * (NG * 64) + lane;
                float vals[3 * NG];
#pragma unroll
                for (int x = 0; x < 3 * NG; ++x) vals[x] = 0.f;
#pragma unroll
                for (int x = 0; x < VEC; ++x) vals[x] = acc[x];
                vals[VEC] = gm; vals[VEC + 1] = den;
#pragma unroll
                for (int k = 0; k < NG; ++k)
                    st_agent16(dst + k * 64, u32x4{__builtin_bit_cast(uint32_t, vals[3 * k]), __builtin_bit_cast(uint32_t, vals[3 * k + 1]),
                                                   __builtin_bit_cast(uint32_t, vals[3 * k + 2]), (uint32_t)share + 1u});
                shstamp<STAMP>(p, wave, 7);
                ++ord;
                continue;
            }
            if (kind == kSegMerge) {
                if (early && cross_left > 0) {                       // walked early, the hand-off is still on its way: finished when it is
                    pend_m = true; pend_b = b; pend_gm = gm; pend_den = den;
#pragma unroll
                    for (int x = 0; x < VEC; ++x) pend_acc[x] = acc[x];
                } else {
                    int spins2 = 0;
                    while (cross_left > 0 && !failed) {             // whatever of the hand-off is still open: now it is waited for
                        cross_step();
                        if (cross_left > 0) __builtin_amdgcn_s_sleep(2);
                        if (++spins2 > kPollLimit) failed = true;
                    }
                    finish_merge(b, gm, den, acc);
                }
            } else {
                store_out(b, den, acc);
            }
#ifdef M355_TAIL_STAMPS
            if (k == nreq - 1) shstamp<STAMP>(p, wave, 7);
#endif
            ++ord;
        }
        if (pend_m) {                                                // every segment is served; the hand-off is all that is left
            int spins2 = 0;
            while (cross_left > 0 && !failed) {
                cross_step();
                if (cross_left > 0) __builtin_amdgcn_s_sleep(2);
                if (++spins2 > kPollLimit) failed = true;
            }
            if (pend_m) finish_merge(pend_b, pend_gm, pend_den, pend_acc);   // (not yet done by the step that completed the hand-off)
        }
        if (failed && lane == 0) __hip_atomic_store(reinterpret_cast<uint32_t*>(p.plan) + kHError, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        wait_vmcnt<0>();                                             // (no LDS-DMA may outlive its wave; none is left unless the hand-off came late)
        if constexpr (STAMP) shstamp<STAMP>(p, wave, 5);
        return;
    }

    // ============================================================================================ streaming wave
    const int j = wave;
    if (j >= NS) { wait_vmcnt<0>(); __syncthreads(); return; }
    const int Nw = (NS - j + W - 1) / W;                                // this wave's stages: j, j + W, ...
    uint8_t* const wbase = smem + j * S::WAVEB;
    uint8_t* const ringp = wbase;
    const int32_t* const recr = reinterpret_cast<const int32_t*>(wbase + 2 * M::STAGEB);
    uint8_t* const qbuf = wbase + 2 * M::STAGEB + 4 * S::RECBYTES;
    const uint32_t ring0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)ringp);
    const uint32_t q0 = rec0 + 4 * S::RECBYTES;

    const int n = lane & 15;                             // q column / K row / d column, by operand
    const int kg = lane >> 4;                            // 16-lane group = k sub-block
    const uint8_t* kbase = reinterpret_cast<const uint8_t*>(p.k_pool) + (int64_t)h * M::ROWB;
    const uint8_t* vbase = reinterpret_cast<const uint8_t*>(p.v_pool) + (int64_t)h * M::ROWB;
    const int64_t row_bytes = p.row_elems * 2;
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
    const bool fused = p.k_new != nullptr;

    f32x4 O[M::DT];
    float m_run = kNegBig, l_run = 0.f;
    bf16x8 qf[M::KS];
    // the fused KV store's pool write waits in registers until the next counted wait has passed (a store in front of
    // it would be waited for as well)
    bool pend = false;
    u32x4 pend_data = {0u, 0u, 0u, 0u};
    uint16_t* pend_dst = nullptr;

    auto rec_dma = [&](const int s) {                    // record of local stage s -> record ring slot s & 3
        const int ss = s < Nw ? s : Nw - 1;              // (past the end: a harmless re-fetch keeps the DMA count uniform)
        if (lane < kRecW) glds4(grec + (int64_t)ss * W * kRecW + lane, rec0 + (s & 3) * S::RECBYTES);
    };
    auto issue = [&](const int s) {
        rec_dma(s + 2);
        const int32_t* rc = recr + (s & 3) * kRecW;
        const int b = __builtin_amdgcn_readfirstlane(rc[kRecB]);
        const int meta = __builtin_amdgcn_readfirstlane(rc[kRecMeta]);
        int64_t ro[M::NI];
        int tt[M::NI];
#pragma unroll
        for (int i = 0; i < M::NI; ++i) {
            tt[i] = i * M::RPI + drow;
            ro[i] = (int64_t)rc[tt[i]] * row_bytes;
        }
        const uint8_t* qsrc = reinterpret_cast<const uint8_t*>(p.q + (int64_t)b * p.q_stride + (int64_t)h * G * D);
#pragma unroll
        for (int k = 0; k < S::NQ; ++k) {
            const int l16 = lane + 64 * k;
            if (l16 < S::QL) glds16(qsrc + l16 * 16, q0 + (s & 1) * S::QBP + k * 1024);
        }
        const uint32_t sbase = ring0 + (s & 1) * M::STAGEB;
        if (fused && (meta & kRecNewest)) {              // wave-uniform: this stage holds the request's newest token
            const int last = (meta & 0xff) - 1;
            const uint8_t* knew = reinterpret_cast<const uint8_t*>(p.k_new + (int64_t)b * p.kn_stride) + h * M::ROWB;
            const uint8_t* vnew = reinterpret_cast<const uint8_t*>(p.v_new + (int64_t)b * p.vn_stride) + h * M::ROWB;
#pragma unroll
            for (int i = 0; i < M::NI; ++i) {
                const bool fresh = tt[i] >= last;        // the newest row (and the padding behind it) comes from k_new / v_new
                glds16p<1>((fresh ? knew : kbase + ro[i]) + koff[i], sbase + 1024 * i);
                glds16p<1>((fresh ? vnew : vbase + ro[i]) + voff[i], sbase + M::TILEB + 1024 * i);
            }
        } else {
#pragma unroll
            for (int i = 0; i < M::NI; ++i) {
                glds16p<1>(kbase + ro[i] + koff[i], sbase + 1024 * i);
                glds16p<1>(vbase + ro[i] + voff[i], sbase + M::TILEB + 1024 * i);
            }
        }
    };
    auto consume = [&](const int s) {
        const int32_t* rc = recr + (s & 3) * kRecW;
        const int meta = __builtin_amdgcn_readfirstlane(rc[kRecMeta]);
        const int valid = meta & 0xff;
        const uint8_t* st = ringp + (s & 1) * M::STAGEB;
        if (meta & kRecFirst) {                          // a new segment for this wave: its q rows, a fresh state
            const uint8_t* qsp = qbuf + (s & 1) * S::QBP;
#pragma unroll
            for (int k = 0; k < M::KS; ++k) {
                u32x4 t = *reinterpret_cast<const u32x4*>(qsp + qa0 + 64 * k);
                if (n >= G) t = u32x4{0u, 0u, 0u, 0u};
                qf[k] = __builtin_bit_cast(bf16x8, t);
            }
            m_run = kNegBig; l_run = 0.f;
#pragma unroll
            for (int dt = 0; dt < M::DT; ++dt) O[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < M::KS; ++k) {
            const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(st + (ka0 ^ (64 * k)));
            const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(st + 16 * M::ROWB + (ka0 ^ (64 * k)));
            s0 = mfma_16x16x32<F16>(a0, qf[k], s0);
            s1 = mfma_16x16x32<F16>(a1, qf[k], s1);
        }
        if (valid < kStageTok) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (4 * kg + r >= valid) s0[r] = kNegBig;
                if (16 + 4 * kg + r >= valid) s1[r] = kNegBig;
            }
        }
        float mx = fmaxf(fmaxf(fmaxf(s0[0], s0[1]), fmaxf(s0[2], s0[3])), fmaxf(fmaxf(s1[0], s1[1]), fmaxf(s1[2], s1[3])));
        float a, b2;
        lane_pair<16>(mx, a, b2);
        mx = fmaxf(a, b2);
        lane_pair<32>(mx, a, b2);
        mx = fmaxf(a, b2);
        const float m_tile = mx * p.scale_log2;
        const float m_new = m_tile > m_run + kLazyMax ? m_tile : m_run;     // lazy running max (decode_mfma.hip)
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
        if (fused && (meta & kRecNewest)) {              // the newest token's K and V rows: LDS -> registers -> (later) pool slot
            const int row = valid - 1;
            if (lane < 2 * M::C) {
                const int c = lane % M::C;
                const bool isv = lane >= M::C;
                const uint8_t* src = st + (isv ? M::TILEB : 0) + row * M::ROWB + 16 * (c ^ (isv ? M::sV(row) : M::sK(row)));
                pend_data = *reinterpret_cast<const u32x4*>(src);
                // (from the bases already in registers: written as a choice between the two kernel ARGUMENTS, hipcc fetched the
                // chosen pointer with a vector load and waited for it with vmcnt(0) - i.e. for the next stage's DMA as well)
                pend_dst = reinterpret_cast<uint16_t*>(const_cast<uint8_t*>(isv ? vbase : kbase) + (int64_t)rc[row] * row_bytes) + c * 8;
            }
            pend = true;
        }
        if (meta & kRecLast) {                           // this wave's part of the segment is done: (m, l, O) -> LDS
            float x, y;
            lane_pair<16>(l_run, x, y);
            float l = x + y;
            lane_pair<32>(l, x, y);
            l = x + y;
            const int ord = __builtin_amdgcn_readfirstlane(rc[kRecOrd]);
            int spins = 0;
            while (lds_peek(ctl + 1) < ord - p.nslot + 1 && ++spins < kPollLimit) __builtin_amdgcn_s_sleep(1);   // slot still being merged
            if (spins >= kPollLimit && lane == 0)               // gave up: the slot is overwritten while it is merged - say so
                __hip_atomic_store(reinterpret_cast<uint32_t*>(p.plan) + kHError, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            float* dst = parts + ((ord % p.nslot) * W + j) * S::PW;
            if (n < G) {
#pragma unroll
                for (int dt = 0; dt < M::DT; ++dt) *reinterpret_cast<f32x4*>(dst + n * D + 16 * dt + 4 * kg) = O[dt];
                if (kg == 0) { dst[G * D + n] = m_run; dst[G * D + G + n] = l; }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_fetch_add(ctl + 2 + ord % p.nslot, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };
    auto flush_pend = [&]() {
        if (pend) {
            if (lane < 2 * M::C) *reinterpret_cast<u32x4*>(pend_dst) = pend_data;
            pend = false;
        }
    };

    // One stage per wave in flight (eight per CU): measured as fast as two, with half the spread between the waves of
    // a workgroup, and every wait is a plain vmcnt(0).  The ring has two slots so that stage s+1 can be asked for as
    // soon as stage s has landed, before it is consumed.
    wait_vmcnt<0>();
    shstamp<STAMP>(p, wave, 2);
    if constexpr (STAMP)                                 // (timing build: which XCD this workgroup runs on, slot 6 of its streaming waves)
        if (p.stamps && lane == 0)
            p.stamps[((size_t)blockIdx.x * (kShareMaxWaves + 1) + wave) * 8 + 6] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15;
    issue(0);
    __syncthreads();                                     // ctl is zero for everyone (nothing reads it before a first flush)
    for (int s = 0; s < Nw; ++s) {
        wait_vmcnt<0>();
        if (s == 0) shstamp<STAMP>(p, wave, 3);
        flush_pend();
        if (s + 1 < Nw) issue(s + 1);
        consume(s);
    }
    shstamp<STAMP>(p, wave, 4);
    flush_pend();
    if constexpr (STAMP) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); shstamp<STAMP>(p, wave, 5); }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
ShareShape share_shape(int Hkv, int D, int G) {
    ShareShape r{0, 0, 0, 0, false};
    if ((D != 64 && D != 128) || (G != 1 && G != 2 && G != 4 && G != 8) || G * D < 64 || Hkv <= 0) return r;
    const int stageb = 128 * D;                          // K + V tile of 32 rows
    const int qb = G * 2 * D, qbp = (qb + 63) / 64 * 64;
    const int waveb = 2 * stageb + 4 * kRecW * 4 + 2 * qbp;
    const int pw = (G * (D + 2) + 3) / 4 * 4;
    constexpr int kLds = 160 * 1024;
    for (int W = share_max_waves(D); W >= 1; --W) {
        for (int ns = 4; ns >= 1; --ns) {
            const int lds = W * waveb + ns * W * pw * 4 + 64 + share_granules(D, G) * 1024;   // ... + ctl + the hand-off staging area
            if (lds <= kLds) {
                r.W = W; r.nslot = ns; r.lds_bytes = lds; r.ok = true;
                r.nb = cu_count() / Hkv;
                if (r.nb < 1) r.nb = 1;
                if (r.nb > 1024) r.nb = 1024;
                return r;
            }
        }
    }
    return r;
}

int launch_share_plan(const int64_t* seq_lens, const int64_t* rows, const int32_t* page_table, int64_t pt_stride, int32_t pt_cols,
                      int32_t B, int32_t ps, int32_t Hkv, int32_t D, int32_t G, int64_t cap, int32_t* plan, hipStream_t st) {
    const ShareShape sh = share_shape(Hkv, D, G);
    if (!sh.ok) return M355_ERR_UNSUPPORTED;
    const ShareLayout L = share_layout(B, sh.nb, Hkv, cap, D, G);
    hipLaunchKernelGGL(share_plan_kernel, dim3(1), dim3(1024), 0, st, seq_lens, B, sh.nb, Hkv, sh.W, cap, (int64_t)pt_cols * ps, plan, L);
    int64_t blocks = (cap + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(share_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, st, page_table, pt_stride, rows, seq_lens, B, ps, sh.W, plan, L);
    return check_launch();
}

template <int D, int G, bool F16>
static int launch_share(const ShareParams& p, int lds, hipStream_t st) {
    // more than 64 KiB of dynamic LDS has to be allowed once per kernel and device
    static std::atomic<uint64_t> allowed{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t bit = 1ull << (dev & 63);
    if (!(allowed.load(std::memory_order_acquire) & bit)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&decode_share_kernel<D, G, F16>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            (void)hipGetLastError();
            return M355_ERR_LAUNCH;
        }
        allowed.fetch_or(bit, std::memory_order_release);
    }
    hipLaunchKernelGGL((decode_share_kernel<D, G, F16>), dim3((unsigned)(p.nb * p.Hkv)), dim3(64 * (p.W + 1)), lds, st,
                       (const int32_t*)p.plan, p.off_rec, p.qcap, p.W, p.Hkv, p.rot, p);
    return M355_OK;
}

int launch_share_attn(ShareParams& p, int D, int G, bool f16, hipStream_t st, bool stamp) {
    const ShareShape sh = share_shape(p.Hkv, D, G);
    if (!sh.ok) return M355_ERR_UNSUPPORTED;
    p.nb = sh.nb; p.W = sh.W; p.nslot = sh.nslot;
    p.rot = sh.nb / p.Hkv > 0 ? sh.nb / p.Hkv : 1;
#ifdef M355_DEV
    if (stamp && !f16) {                                   // dev timing builds: the benchmark shape, the 8B and the 70B/TP8 per-rank shape
#define M355_STAMPED(DD, GG) if (D == DD && G == GG) { \
            static bool allowed = false; \
            if (!allowed) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&decode_share_kernel<DD, GG, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); allowed = true; } \
            hipLaunchKernelGGL((decode_share_kernel<DD, GG, false, true>), dim3((unsigned)(p.nb * p.Hkv)), dim3(64 * (p.W + 1)), sh.lds_bytes, st, \
                               (const int32_t*)p.plan, p.off_rec, p.qcap, p.W, p.Hkv, p.rot, p); \
            return M355_OK; }
        M355_STAMPED(64, 4) M355_STAMPED(128, 4) M355_STAMPED(128, 8)
#undef M355_STAMPED
    }
#else
    (void)stamp;
#endif
#define M355_CASE(DD, GG) if (D == DD && G == GG) return f16 ? launch_share<DD, GG, true>(p, sh.lds_bytes, st) : launch_share<DD, GG, false>(p, sh.lds_bytes, st);
    M355_CASE(64, 1) M355_CASE(64, 2) M355_CASE(64, 4) M355_CASE(64, 8)
    M355_CASE(128, 1) M355_CASE(128, 2) M355_CASE(128, 4) M355_CASE(128, 8)
#undef M355_CASE
    return M355_ERR_UNSUPPORTED;
}

}  // namespace m355
