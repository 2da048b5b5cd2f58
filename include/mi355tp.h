/*
 * mi355tp.h - C ABI of the tensor-parallel epilogue in libmi355attn.so: the step right AFTER the
 * attention path, o_proj as a row-parallel linear layer followed by a SUM all-reduce.
 *
 * Reference: RowParallelLinear.forward (layers/linear.py:228-232) computes `F.linear(x, w)` and then
 * `dist.all_reduce(out)`; its caller is the attention block (models/llama.py:137).  At decode sizes ([B <= 256, hidden]
 * bf16 = 512 KiB - 2 MiB) the collective is latency-bound, so here the two are ONE launch per rank:
 *
 *   every rank  computes its partial  P_r = x_r [T, K] . w_r [N, K]^T   (fp32 accumulate on the matrix cores, rounded
 *               once to the element type, as F.linear does),
 *               PUSHES each 16 x 64 tile of P_r into slot r of every rank's receive buffer through xGMI peer
 *               pointers (all links of the GPU at once) and raises a per-tile flag on each peer,
 *               waits for the same tile of every other rank, adds the `world` slots IN RANK ORDER in fp32 and
 *               writes out[T, N]  - every rank adds the same numbers in the same order, so all ranks hold bit-identical
 *               results (a ring all-reduce gives every rank the same bits too; TP needs that).
 *
 * Set-up (once): each rank allocates one receive buffer and one flag array with m355_comm_alloc (uncached device
 * memory, so data written by a peer is never served from a stale cache line), exports them (m355_comm_export, a 64-byte
 * handle that travels by any host channel), opens the peers' (m355_comm_open).  Per call nothing is allocated and
 * nothing synchronises; the epoch must grow by one per call on every rank (flags are never cleared; the receive buffer
 * has two halves used by even / odd epochs, which is enough because a rank cannot begin call e+2 before every peer has
 * finished reading call e).  Waits are bounded: a peer that never arrives raises *error (device int32) instead of
 * hanging the GPU.
 *
 * Conventions as in mi355attn.h (device pointers, strides in elements, `stream` a hipStream_t, int return codes).
 */
#ifndef MI355TP_H
#define MI355TP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define M355_MAX_RANKS 8

/* One rank's view of a fused o_proj call.  recv[p] / flags[p] are rank p's receive buffer / flag array as addressable
 * from THIS rank's device (own allocation for p == rank, an opened handle otherwise). */
typedef struct M355OprojRank {
    const void *x;            /* [T, K]  this rank's attention output (its heads' columns)           */
    int64_t x_row_stride;
    const void *w;            /* [N, K]  this rank's column slice of the o_proj weight               */
    int64_t w_row_stride;
    void *out;                /* [T, N]  the reduced result                                          */
    int64_t out_row_stride;
    void *recv[M355_MAX_RANKS];
    uint32_t *flags[M355_MAX_RANKS];
    int32_t rank;
    int32_t reserved;
} M355OprojRank;

/* Uncached, zero-filled device memory on the current device / its release.  Set-up time only (both synchronise the device).
 * m355_comm_free does NOT hand the memory back to the driver: the block goes to a pool inside the library and the next
 * m355_comm_alloc of the same size on the same device gets it back, zeroed.  (On ROCm 7.2 / gfx950 a normal allocation
 * that later lands on the address range of a hipFree'd uncached one was seen read through stale 4 KiB translations - whole
 * pages of a fresh tensor holding old data - so uncached memory stays with the library until the process ends.)
 * m355_comm_free of a pointer m355_comm_alloc did not return: M355_ERR_INVALID_ARG. */
int m355_comm_alloc(int64_t bytes, void **ptr);
int m355_comm_free(void *ptr);
/* Zero an allocation again (set-up time; synchronises): for a caller that keeps buffers and re-uses them. */
int m355_comm_zero(void *ptr, int64_t bytes);
/* Inter-process handle (64 bytes, host memory) of a m355_comm_alloc allocation; open / close it in another process. */
int m355_comm_export(void *ptr, void *handle64);
int m355_comm_open(const void *handle64, void **ptr);
int m355_comm_close(void *ptr);

/* Bytes of one rank's receive buffer (2 halves x world slots x T_cap x N elements) and flag array. */
int64_t m355_oproj_recv_bytes(int32_t world, int32_t T_cap, int32_t N);
int64_t m355_oproj_flag_bytes(int32_t world, int32_t T_cap, int32_t N);

/* out = sum over ranks of x_r . w_r^T, replacing F.linear + dist.all_reduce (layers/linear.py:228-232).
 * `ranks` (HOST memory) holds n_ranks entries: 1 in production (this process's rank); n_ranks == world runs every rank
 * of a world whose buffers all live on THIS device in one launch (how the single-GPU parity test drives it).
 * T <= T_cap rows; N % 64 == 0; K % 32 == 0; bf16 or f16.  error: device int32 raised to 1 when a wait gave up, or NULL.
 * A wait that gives up (a peer did not arrive within ~1 s of polls) also writes NaN into that tile's rows of `out`, so the
 * result cannot pass for a sum; from then on the instance is OUT OF STEP with its peers and unusable - a late push may
 * land in a half that a faster rank has re-used - until every rank has zeroed its receive buffer, flag array and
 * epoch_state again (m355_comm_zero) and the host epoch restarts.  Check *error wherever the caller synchronises anyway.
 * T == 0 launches nothing and is NOT a call: with a host-counted epoch do not advance it (the two buffer halves rely on
 * a completed call of the other parity in between).
 * n_ranks == world > 1 (all ranks in one launch on one device, the parity test's form) needs every workgroup of the
 * launch resident at once: ceil(T/16) * N/64 * world beyond what the device holds is refused with M355_ERR_INVALID_ARG.
 * The call's epoch is `epoch` (counted by the host: 1, 2, 3, ...) when epoch_state is NULL.  Otherwise epoch_state is
 * device memory of m355_oproj_flag_bytes(world, T_cap, N) bytes, zeroed once at set-up, where the launch keeps a call
 * count per (rank, tile) and advances it itself: the SAME launch (same arguments) can then be replayed from a captured HIP
 * graph; `epoch` is ignored.  Every rank must use the same form, and with it the same sequence of calls. */
int m355_oproj_allreduce(const M355OprojRank *ranks, int32_t n_ranks, int32_t world, int32_t T, int32_t N, int32_t K,
                         int32_t T_cap, uint32_t epoch, uint32_t *epoch_state, int32_t *error, int32_t dtype, void *stream);

#ifdef __cplusplus
}
#endif
#endif
