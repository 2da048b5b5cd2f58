// Declarations shared by the prefill (extend) attention kernels: prefill_attn.hip (32 query rows per wave, two
// workgroups per CU) and prefill_wg2.hip (64 rows per wave, one workgroup per CU).
#pragma once
#include "common.h"

#ifndef M355_PREFILL_POLICY
#define M355_PREFILL_POLICY 0     // K/V tiles are re-read by the row blocks of the same XCD through L2: default policy
#endif
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
    unsigned long long* stamps;                       // dev-only: per-wave phase cycle sums (STAMP builds)
    float* ws;                                        // split-KV partials [T*Hq rows][splits][D + 4: O, m, l, pad] (splits > 1)
    int32_t splits;                                   // KV splits per row block (blockIdx.y)
    int32_t f16;                                      // elements are IEEE half instead of bf16
};

template <int D>
struct WgCfg {
    static constexpr int C = D / 8;                   // 16-byte chunks per row
    static constexpr int ROWB = 2 * D;
    static constexpr int RPI = 64 / C;                // rows per DMA instruction
    static constexpr int NI = 32 / RPI;               // DMA instructions per K (or V) tile
    static constexpr int TILEB = 32 * ROWB;
    static constexpr int BUFB = 2 * TILEB;            // K tile then V tile
    static constexpr int KS = D / 16, DT = D / 32;
    static constexpr int SPAN = 16 / C;               // rows per 256-byte bank span
    // chunk ch of row `row` sits at position ch ^ s(row): conflict-free 16-row ds_read_b128 column reads of K,
    // conflict-free 4-row x 64-byte transposed reads of V
    __device__ static constexpr int sK(int row) { return (row / SPAN) % C; }
    __device__ static constexpr int sV(int row) { return ((row / SPAN) % (C / 4)) * 4; }
};

template <int V> struct IntC { static constexpr int value = V; };
#ifndef M355_PREFILL_PRIO
#define M355_PREFILL_PRIO 2
#endif
constexpr int kPrio = M355_PREFILL_PRIO;          // s_setprio 1 around the MFMA clusters: 1 = QK^T only, 2 = PV too (each +1-2 % on cfg3, A/B on one box)
constexpr float kLazy = 8.f;                          // log2 units
constexpr int kSlotWin = 1024;                        // tokens whose pool slots are staged in LDS at a time


// 64 rows per wave (prefill_wg2.hip); false when the head shape has no such kernel
bool launch_prefill_wg2(const PrefillParams& p, int D, int G, int max_extend_len, hipStream_t st);

}  // namespace m355
