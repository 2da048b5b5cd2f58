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

template <int D, int G, int U>
__global__ __launch_bounds__(64) void decode_split_kernel(const DecodeParams p) {
    constexpr int LPH = D / 8;        // lanes per (token, head) row
    constexpr int NT = kWave / LPH;   // token rows per wave-wide load
    constexpr int TPI = NT * U;       // tokens per iteration
    __shared__ int32_t pages[kMaxChunk];

    const int lane = threadIdx.x;
    const int dchunk = lane % LPH;
    const int tsub = lane / LPH;

    uint32_t bid = blockIdx.x;
    const int split = bid % p.max_splits;  bid /= p.max_splits;
    const int h = bid % p.Hkv;
    const int qi = bid / p.Hkv;

    int64_t L, row;
    query_row_len(p, qi, row, L);
    const int64_t t0 = (int64_t)split * p.chunk;
    if (t0 >= L) return;                                  // wave-uniform
    const int ntok = (int)((L - t0 < p.chunk) ? (L - t0) : p.chunk);

    // ---- stage the chunk's page-table entries in LDS
    const int ps = p.ps, sh = p.ps_shift;
    const int64_t first_pg = sh >= 0 ? (t0 >> sh) : (t0 / ps);
    const int64_t last_tok = t0 + ntok - 1;
    const int npg = (int)((sh >= 0 ? (last_tok >> sh) : (last_tok / ps)) - first_pg) + 1;
    const int32_t* pt_row = p.page_table + row * p.pt_stride + first_pg;
    for (int i = lane; i < npg; i += kWave) pages[i] = pt_row[i];
    __syncthreads();

    // ---- q: the G heads of this kv group, this lane's 8 head-dim elements, packed bf16 pairs
    uint32_t qreg[G][4];
    {
        const uint16_t* qp = p.q + (int64_t)qi * p.q_stride + (int64_t)h * G * D + dchunk * 8;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const u32x4 t = *reinterpret_cast<const u32x4*>(qp + g * D);
            qreg[g][0] = t[0]; qreg[g][1] = t[1]; qreg[g][2] = t[2]; qreg[g][3] = t[3];
        }
    }

    float m[G], l[G], o[G][8];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        m[g] = kNegBig; l[g] = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[g][j] = 0.f;
    }

    const int64_t lane_off = (int64_t)h * D + dchunk * 8;
    const int t0_in_pg = (int)(t0 - (sh >= 0 ? (first_pg << sh) : first_pg * ps));

    auto load_tile = [&](int it, u32x4 (&kt)[U], u32x4 (&vt)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int t = (it * U + u) * NT + tsub;            // token index inside the chunk
            t = t < ntok ? t : ntok - 1;                  // clamp: masked later, address stays valid
            const int tp = t + t0_in_pg;                  // offset from the first staged page's first token
            const int pi = sh >= 0 ? (tp >> sh) : (tp / ps);
            const int in_pg = tp - (sh >= 0 ? (pi << sh) : pi * ps);
            const int64_t slot = (int64_t)pages[pi] * ps + in_pg;
            const int64_t off = slot * p.row_elems + lane_off;
            kt[u] = *reinterpret_cast<const u32x4*>(p.k_pool + off);
            vt[u] = *reinterpret_cast<const u32x4*>(p.v_pool + off);
        }
    };

    auto compute_tile = [&](int it, const u32x4 (&kt)[U], const u32x4 (&vt)[U]) {
        float s[U][G];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int g = 0; g < G; ++g) {
                float a = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) a = dot2_bf16(qreg[g][w], kt[u][w], a);
                s[u][g] = a;
            }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int g = 0; g < G; ++g) s[u][g] = group_sum<LPH>(s[u][g]);

        bool valid[U];
#pragma unroll
        for (int u = 0; u < U; ++u) valid[u] = ((it * U + u) * NT + tsub) < ntok;

        float pw[U][G], alpha[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float mx = m[g];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                s[u][g] = valid[u] ? s[u][g] * p.scale_log2 : kNegBig;
                mx = fmaxf(mx, s[u][g]);
            }
            alpha[g] = exp2_fast(m[g] - mx);
            m[g] = mx;
            float sum = 0.f;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float e = valid[u] ? exp2_fast(s[u][g] - mx) : 0.f;
                pw[u][g] = e;
                sum += e;
            }
            l[g] = l[g] * alpha[g] + sum;
        }
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int j = 0; j < 8; ++j) o[g][j] *= alpha[g];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float vf[8];
#pragma unroll
            for (int w = 0; w < 4; ++w) { vf[2 * w] = bf16_lo(vt[u][w]); vf[2 * w + 1] = bf16_hi(vt[u][w]); }
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int j = 0; j < 8; ++j) o[g][j] = fmaf(pw[u][g], vf[j], o[g][j]);
        }
    };

    const int n_it = (ntok + TPI - 1) / TPI;
    u32x4 ka[U], va[U], kb[U], vb[U];
    load_tile(0, ka, va);
    int it = 0;
    // two-stage software pipeline, unrolled by two so both buffers have static names
    for (; it + 1 < n_it; it += 2) {
        load_tile(it + 1, kb, vb);
        compute_tile(it, ka, va);
        if (it + 2 < n_it) load_tile(it + 2, ka, va);
        compute_tile(it + 1, kb, vb);
    }
    if (it < n_it) compute_tile(it, ka, va);

    // ---- merge the NT per-lane-group states of this wave (lane bits above the dchunk bits)
#pragma unroll
    for (int mask = LPH; mask < kWave; mask <<= 1) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const float m2 = __shfl_xor(m[g], mask);
            const float l2 = __shfl_xor(l[g], mask);
            const float mx = fmaxf(m[g], m2);
            const float wa = exp2_fast(m[g] - mx), wb = exp2_fast(m2 - mx);
            l[g] = l[g] * wa + l2 * wb;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[g][j] = o[g][j] * wa + __shfl_xor(o[g][j], mask) * wb;
            m[g] = mx;
        }
    }

    if (tsub != 0) return;
    const int nsplit = (int)((L + p.chunk - 1) / p.chunk);
    if (nsplit == 1) {
        uint16_t* op = p.out + (int64_t)qi * p.o_stride + (int64_t)h * G * D + dchunk * 8;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const float inv = 1.f / l[g];
            u32x4 w;
#pragma unroll
            for (int x = 0; x < 4; ++x) w[x] = pack_bf16(o[g][2 * x] * inv, o[g][2 * x + 1] * inv);
            *reinterpret_cast<u32x4*>(op + g * D) = w;
        }
    } else {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int64_t idx = ((int64_t)qi * p.Hq + h * G + g) * p.max_splits + split;
            float* wo = p.ws_o + idx * D + dchunk * 8;
            *reinterpret_cast<f32x4*>(wo) = f32x4{o[g][0], o[g][1], o[g][2], o[g][3]};
            *reinterpret_cast<f32x4*>(wo + 4) = f32x4{o[g][4], o[g][5], o[g][6], o[g][7]};
            if (dchunk == 0) { p.ws_ml[idx * 2] = m[g]; p.ws_ml[idx * 2 + 1] = l[g]; }
        }
    }
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

template <int D, int G>
static void launch_decode(const DecodeParams& p, hipStream_t st) {
    constexpr int U = (G >= 8) ? 2 : 4;
    const uint32_t blocks = (uint32_t)((int64_t)p.NQ * p.Hkv * p.max_splits);
    hipLaunchKernelGGL((decode_split_kernel<D, G, U>), dim3(blocks), dim3(64), 0, st, p);
    if (p.max_splits > 1)
        hipLaunchKernelGGL((decode_merge_kernel<D>), dim3((uint32_t)(p.NQ * p.Hq)), dim3(64), 0, st, p);
}

// Shared by m355_decode_attn (cu_q == null: one query per request) and by the row-parallel extend path of
// m355_prefill_attn (cu_q/prefix_lens given: NQ packed query rows, each treated as its own decode query).
int paged_rows_attn(const void* q, int64_t q_stride, const void* k_pool, const void* v_pool, const int32_t* page_table,
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
    p.q_stride = q_stride; p.o_stride = o_stride; p.pt_stride = pt_stride; p.row_elems = (int64_t)Hkv * D;
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
#define M355_CASE(DD, GG) if (D == DD && G == GG) { launch_decode<DD, GG>(p, st); return check_launch(); }
    M355_CASE(64, 1) M355_CASE(64, 2) M355_CASE(64, 4) M355_CASE(64, 8)
    M355_CASE(128, 1) M355_CASE(128, 2) M355_CASE(128, 4) M355_CASE(128, 8)
#undef M355_CASE
    return M355_ERR_UNSUPPORTED;
}

int32_t decode_pick_chunk(int64_t nq, int32_t Hkv, int32_t max_len, int32_t chunk) {
    return chunk > 0 ? chunk : auto_chunk(nq, Hkv, max_len);
}

}  // namespace m355

extern "C" int64_t m355_decode_workspace_bytes(int32_t B, int32_t Hq, int32_t D, int32_t max_seq_len_hint,
                                               int32_t chunk_tokens) {
    if (B <= 0 || Hq <= 0 || D <= 0 || max_seq_len_hint <= 0) return 0;
    // Hkv is not known here; size for the smallest chunk the auto rule can pick unless one is given
    const int32_t chunk = chunk_tokens > 0 ? chunk_tokens : 64;
    const int64_t splits = (max_seq_len_hint + chunk - 1) / chunk;
    if (splits <= 1) return 0;
    return (int64_t)B * Hq * splits * (D + 2) * (int64_t)sizeof(float);
}

extern "C" int m355_decode_attn(const void* q, int64_t q_row_stride, const void* k_pool, const void* v_pool,
                                const int32_t* page_table, int64_t pt_row_stride, const int64_t* page_table_ids,
                                const int64_t* seq_lens, void* out, int64_t o_row_stride, int32_t B, int32_t Hq,
                                int32_t Hkv, int32_t D, int32_t page_size, float scale, int32_t max_seq_len_hint,
                                int32_t chunk_tokens, void* workspace, int64_t workspace_bytes, int32_t dtype,
                                void* stream) {
    return m355::paged_rows_attn(q, q_row_stride, k_pool, v_pool, page_table, pt_row_stride, page_table_ids, seq_lens,
                                 nullptr, nullptr, B, out, o_row_stride, B, Hq, Hkv, D, page_size, scale,
                                 max_seq_len_hint, chunk_tokens, workspace, workspace_bytes, dtype, (hipStream_t)stream);
}
