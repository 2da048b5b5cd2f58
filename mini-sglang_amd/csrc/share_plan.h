// Plan of the single-launch decode kernel (decode_share.hip): the batch's token stream cut into one SHARE per
// workgroup, every share expanded into 32-token STAGES whose records a wave fetches by LDS-DMA as it goes.
#pragma once
#include "stream_plan.h"

namespace m355 {

constexpr int kStageTok = 32;                   // tokens per stage (one K tile + one V tile of one kv head)
constexpr int kRecW = 40;                       // int32 words per stage record
constexpr int kRecB = 32;                       // record word: request index
constexpr int kRecMeta = 33;                    // record word: valid tokens (low 8 bits) | flags
constexpr int kRecOrd = 34;                     // record word: ordinal of the stage's segment inside its share
constexpr int kRecFirst = 1 << 8;               // meta flag: the owning wave's first stage of this segment
constexpr int kRecLast = 1 << 9;                // meta flag: the owning wave's last stage of this segment
constexpr int kRecNewest = 1 << 10;             // meta flag: token valid-1 of the stage is the request's newest token
constexpr int kShareHdr = 16;                   // plan header words
constexpr int kShareW = 8;                      // words per share entry: {first request, last request, first request walked second, ...}
constexpr int kShareEarlyStages = 8;            // a first request of at most this many stages (one round of the 8 waves) is walked second
constexpr int kSegWhole = 0, kSegPublish = 1, kSegMerge = 2;     // a segment = one request's stages inside one share
constexpr int kShareMinStages = 4;              // shortest share
constexpr int kShareMaxWaves = 8;               // streaming waves per workgroup (plus one finishing wave)
constexpr int kShareCumLds = 4096;              // requests whose prefix sums the plan kernels stage in LDS

// header words
constexpr int kHTotal = 0, kHShareStages = 1, kHB = 2, kHShares = 3, kHWaves = 4, kHStages = 5, kHError = M355_PLAN_ERROR_WORD, kHShareRem = 7;

// hand-off granules per (kv head, share): one fp32 partial O[G][D] + m, l per lane, as 16-byte granules {3 values, tag}
__host__ __device__ constexpr int share_granules(int D, int G) { return ((G * D / 64 > 0 ? G * D / 64 : 1) + 2 + 2) / 3; }
__host__ __device__ inline int64_t share_gran_words(int D, int G) { return (int64_t)share_granules(D, G) * 64 * 4; }

// plan buffer (int32 words): hdr | cum[B+1] | stg[B+1] | nz[B+1] | share[nb][kShareW] | gran[Hkv*nb][...] | rec[nb][qcap][kRecW]
// Share i's records start at a place that depends on i alone (i * qcap), so a wave asks for its first records with its
// very first instructions, beside the header loads that tell it how many stages the share really has.
struct ShareLayout {
    int64_t cum, stg, nz, share, gran, gran_words, rec, ints;
    int64_t max_stages;
    int32_t qcap;                                   // record places per share (>= the stages any share can get)
};
__host__ __device__ inline ShareLayout share_layout(int32_t B, int32_t nb, int32_t Hkv, int64_t cap, int32_t D, int32_t G) {
    ShareLayout L;
    L.max_stages = cap / kStageTok + B + 2;         // every request adds at most one partial stage
    L.cum = kShareHdr;
    L.stg = L.cum + (B + 1);
    L.nz = L.stg + (B + 1);
    L.share = (L.nz + (B + 1) + 3) & ~(int64_t)3;
    L.gran = (L.share + (int64_t)nb * kShareW + 63) & ~(int64_t)63;
    L.gran_words = (int64_t)Hkv * nb * share_gran_words(D, G);
    L.rec = (L.gran + L.gran_words + 63) & ~(int64_t)63;                   // records start on a 256-byte line
    int64_t q = (L.max_stages + nb - 1) / nb;
    L.qcap = (int32_t)(q < kShareMinStages ? kShareMinStages : q);
    L.ints = L.rec + ((int64_t)nb * L.qcap + 2 * kShareMaxWaves) * kRecW;  // (+ slack: a wave's first two record fetches are unconditional)
    return L;
}

// Launch shape of the share kernel for a head shape: W streaming waves per workgroup, NSLOT partial slots, LDS bytes.
struct ShareShape {
    int W, nslot, lds_bytes, nb;     // nb: shares (= workgroups per kv head)
    bool ok;
};
ShareShape share_shape(int Hkv, int D, int G);

struct ShareParams {
    const uint16_t* q;
    uint16_t* k_pool;             // written only by the fused KV store
    uint16_t* v_pool;
    const uint16_t* k_new;        // fused KV store: the step's new K/V rows [B, Hkv*D], or nullptr
    const uint16_t* v_new;
    int64_t kn_stride, vn_stride;
    int32_t* plan;                // the kernel raises and lowers the hand-off granules inside it
    uint16_t* out;
    int64_t q_stride, o_stride, row_elems;
    int64_t off_stg, off_share, off_gran, off_rec;
    int32_t B, Hq, Hkv, nb, W, nslot, rot, qcap;
    float scale_log2;
    unsigned long long* stamps;   // dev-only timing probe (tools/stamp_share.py), 8 slots per wave; else nullptr
};

int launch_share_plan(const int64_t* seq_lens, const int64_t* rows, const int32_t* page_table, int64_t pt_stride, int32_t pt_cols,
                      int32_t B, int32_t ps, int32_t Hkv, int32_t D, int32_t G, int64_t cap, int32_t* plan, hipStream_t st);
int launch_share_attn(ShareParams& p, int D, int G, bool f16, hipStream_t st, bool stamp = false);

}  // namespace m355
