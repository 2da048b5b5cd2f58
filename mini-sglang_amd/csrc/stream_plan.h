// Shared by the planned decode kernels (decode_stream.hip: plan + VALU attention, decode_mfma.hip: MFMA attention).
#pragma once
#include "common.h"

namespace m355 {

constexpr int kRMax = 1024;                     // tokens per range (LDS slot staging: 4 KiB)
constexpr int kMaxPieces = 8;                   // pieces a range record holds (more: the range is walked by scanning cum)
constexpr int kRecEnds = 4 + 4 * kMaxPieces;    // record word: bit k set = piece k ends its request
constexpr int kRecCut = kRecEnds + 1;           // 2 words per piece: {pieces its request is cut into (0: the piece is a whole request),
                                                //                     ordinal of this piece among them}
constexpr int kRec = (kRecCut + 2 * kMaxPieces + 3) / 4 * 4;   // ints per record: n, np|-1, b0, g0, pieces{b,start,len,pidx}, ends, cuts
static_assert(kRec <= 64, "a wave fetches a record with one load per lane");
constexpr int kSpec = 5;                        // slots loaded per lane together with the record (5*64 = 320 tokens)
constexpr int kHdrInts = 8;

constexpr float kLazyMax = 8.f;                 // log2 units: the running max moves only when beaten by 2^8

// Geometry of one 32-token K (or V) tile of one kv head in LDS, shared by the matrix-core decode kernels.
template <int D>
struct MG {
    static constexpr int C = D / 8;             // 16-byte chunks per K/V row of one head
    static constexpr int ROWB = 2 * D;          // bytes per row
    static constexpr int RPI = 64 / C;          // rows one DMA instruction brings (1 KiB)
    static constexpr int NI = 32 / RPI;         // DMA instructions per 32-row tile
    static constexpr int TILEB = 32 * ROWB;     // K (or V) tile bytes
    static constexpr int STAGEB = 2 * TILEB;
    static constexpr int KS = D / 32;           // k-steps of S^T = K Q^T
    static constexpr int DT = D / 16;           // d-tiles of O^T
    static constexpr int SPAN = 16 / C;         // rows per 256-byte bank span
    // 16-byte chunk ch of row `row` lives at chunk position ch ^ s(row) of that row: conflict-free ds_read_b128 row
    // reads of K (16 rows, one chunk column) and conflict-free transposed reads of V (8 rows x 2 chunks per half wave)
    __host__ __device__ static constexpr int sK(int row) { return (row / SPAN) % C; }
    __host__ __device__ static constexpr int sV(int row) { return ((row / SPAN) % (C / 2)) * 2; }
};

// plan buffer (int32 words): hdr[8] | cum[B+1] | pbase[B+1] | full[B+1] | arrivals[max_ranges + B][Hkv] | rec[max_ranges][kRec] |
//                            slots[max_ranges][kRMax]
// full[b] = 1 when the whole of request b is in the stream (0: cut by the capacity or by the page table's width - its
// newest token then has no slot the fused store could write).  hdr[M355_PLAN_ERROR_WORD] stays 0: this form never waits.
// arrivals: one count per (partial, kv head), zero between launches (the plan kernel clears them, the wave that completes
// a count lowers it again): the pieces of a cut request count themselves in, the last one to arrive merges (decode_mfma.hip).
struct PlanLayout {
    int64_t cum, pbase, full, arrivals, rec, slots, ints;
    int32_t max_ranges;
};
__host__ __device__ inline PlanLayout plan_layout(int32_t B, int32_t nwh, int32_t Hkv, int64_t cap) {
    PlanLayout L;
    L.max_ranges = (int32_t)(nwh + cap / kRMax + 2);
    L.cum = kHdrInts;
    L.pbase = L.cum + (B + 1);
    L.full = L.pbase + (B + 1);
    L.arrivals = (L.full + (B + 1) + 3) & ~(int64_t)3;
    L.rec = (L.arrivals + ((int64_t)L.max_ranges + B) * Hkv + 3) & ~(int64_t)3;
    L.slots = L.rec + (int64_t)L.max_ranges * kRec;
    L.ints = L.slots + (int64_t)L.max_ranges * kRMax;
    return L;
}

// Merging the partials of a request cut into np pieces: up to kMergeFlat by the one wave that arrives last; more in two
// levels - groups of merge_group(np) consecutive pieces (the group's last arriver leaves ONE partial in the place of the
// group's first piece), then at most 16 group partials by the last group to finish.
constexpr int kMergeFlat = 24;
__host__ __device__ inline int32_t merge_group(int32_t np) { return np <= kMergeFlat ? np : (np <= 256 ? 16 : (np + 15) / 16); }

struct StreamParams {
    const uint16_t* q;
    const uint16_t* k_pool;
    const uint16_t* v_pool;
    const uint16_t* k_new;        // fused KV store (MFMA kernel): the step's new K/V rows [B, Hkv*D], or nullptr
    const uint16_t* v_new;
    int64_t kn_stride, vn_stride;
    const int32_t* plan;
    uint16_t* out;
    float* ws_o;    // [n_partials, Hq, D]
    float* ws_ml;   // [n_partials, Hq, 2]
    int64_t q_stride, o_stride, row_elems;
    int64_t off_cum, off_pbase, off_full, off_arrivals, off_rec, off_slots;
    int32_t B, Hq, Hkv, nwh, max_ranges;
    int32_t merge_in_kernel;      // 1: the last piece of a cut request to arrive merges (product); 0: stream_merge_kernel does (dev A/B)
    int32_t rot;                  // kv head of workgroup (w, j) = (j + w / rot) % Hkv; 0 = no rotation
    float scale_log2;
    unsigned long long* stamps;   // dev-only timing probe (tools/stamp_decode.py), 8 slots per wave; else nullptr
};

// s_memrealtime per wave at phase boundaries when the STAMP build of a kernel runs with a stamp buffer
template <bool ON>
__device__ __forceinline__ void sstamp(const StreamParams& p, int k) {
    if constexpr (ON)
        if (p.stamps && threadIdx.x == 0) p.stamps[(size_t)blockIdx.x * 8 + k] = __builtin_amdgcn_s_memrealtime();
}



// MFMA attention kernel (decode_mfma.hip).  Its waves-per-CU is LDS-limited and depends on the token capacity.
int mfma_waves_per_cu(int Hkv, int D, int G, int64_t cap);
int mfma_rmin();                 // shortest range (tokens) the plan should make for it
int launch_stream_mfma(const StreamParams& p, int D, int G, int64_t cap, hipStream_t st, bool stamp, bool f16);

}  // namespace m355
