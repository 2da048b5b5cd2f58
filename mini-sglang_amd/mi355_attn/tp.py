"""Tensor-parallel head sharding of the attention path.

The reference shards attention by heads with no communication inside the backend: rank r of tp owns the
contiguous q-heads [r*Hq/tp, (r+1)*Hq/tp) and kv-heads [r*Hkv/tp, (r+1)*Hkv/tp) (models/llama.py:76-87,
layers/linear.py:171-190), keeps a KV pool holding only its kv-heads (engine/model_runner.py:161) and an
identical page table on every rank.  The one exchange step is the SUM all-reduce RowParallelLinear issues on
the o_proj output (layers/linear.py:228-232) - RCCL over xGMI on the GPUs ("nccl" backend), gloo in CPU tests.
"""
from __future__ import annotations

import ctypes
import dataclasses
from typing import List, Optional

import torch
import torch.distributed as dist


@dataclasses.dataclass(frozen=True)
class HeadShard:
    tp_size: int
    tp_rank: int
    num_heads: int          # per-rank q heads
    num_kv_heads: int       # per-rank kv heads
    q_head_start: int       # first global q head owned
    kv_head_start: int      # first global kv head owned
    head_dim: int

    @property
    def group_size(self) -> int:
        return self.num_heads // self.num_kv_heads

    def q_columns(self) -> slice:
        """Columns of the full [T, Hq_total*D] q (and of the attention output) this rank owns."""
        return slice(self.q_head_start * self.head_dim, (self.q_head_start + self.num_heads) * self.head_dim)

    def kv_columns(self) -> slice:
        """Columns of the full [T, Hkv_total*D] k / v this rank owns."""
        return slice(self.kv_head_start * self.head_dim, (self.kv_head_start + self.num_kv_heads) * self.head_dim)


def shard_heads(total_num_heads: int, total_num_kv_heads: int, head_dim: int, tp_size: int, tp_rank: int) -> HeadShard:
    """Per-rank head counts as LlamaAttention computes them (models/llama.py:76-87).  tp_size > kv heads (KV
    replication) is refused: the reference's loader mis-slices it (layers/linear.py:189) and no config uses it."""
    if not 0 <= tp_rank < tp_size:
        raise ValueError(f"tp_rank {tp_rank} outside [0, {tp_size})")
    if total_num_heads % tp_size or total_num_kv_heads % tp_size:
        raise ValueError(f"heads ({total_num_heads}, {total_num_kv_heads}) do not divide over tp_size {tp_size}")
    hq, hkv = total_num_heads // tp_size, total_num_kv_heads // tp_size
    if hq % hkv:
        raise ValueError("a GQA group would straddle ranks")
    return HeadShard(tp_size, tp_rank, hq, hkv, tp_rank * hq, tp_rank * hkv, head_dim)


def all_reduce_rows(x: torch.Tensor, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """In-place SUM all-reduce of a row-parallel partial result (layers/linear.py:228-232); a no-op without an
    initialised process group or with a single rank."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(x, op=dist.ReduceOp.SUM, group=group)
    return x


def row_parallel_linear(x: torch.Tensor, w: torch.Tensor, chunks: int = 1,
                        group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """o_proj / down_proj under TP (first cut): y = all_reduce_sum(x @ w.T), the pair
    RowParallelLinear.forward issues back to back (layers/linear.py:228-232).

    `chunks > 1` pipelines the two over row blocks of x: block i's all-reduce is handed to the collective's own stream
    (`async_op=True`: RCCL orders it after the GEMM that produced the block) while the GEMM of block i+1 runs on the
    caller's stream, so for prefill-size messages the xGMI transfer hides behind the matrix cores.  Every row is
    reduced exactly once, by the same collective over the same ranks, so the result equals the unchunked one up to
    the GEMM's own blocking.  Decode-size messages are latency-bound - chunking only adds collectives there - so
    callers keep chunks=1 below a few thousand rows (bench.py N>1 reports both, `o_proj_all_reduce`)."""
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    T = x.shape[0]
    chunks = max(1, min(int(chunks), T))
    if not multi or chunks == 1:
        return all_reduce_rows(torch.nn.functional.linear(x, w), group)
    y = torch.empty((T, w.shape[0]), dtype=x.dtype, device=x.device)
    step = -(-T // chunks)
    pending = []
    for lo in range(0, T, step):
        blk = y[lo:lo + step]                                  # contiguous row block of the result: reduced in place
        torch.matmul(x[lo:lo + step], w.t(), out=blk)
        pending.append(dist.all_reduce(blk, op=dist.ReduceOp.SUM, group=group, async_op=True))
    for work in pending:
        work.wait()                                            # the caller's stream waits; the host does not block on RCCL
    return y


class FusedRowParallel:
    """o_proj + SUM all-reduce as ONE launch per rank (include/mi355tp.h): replaces the pair
    `F.linear(x, w)`; `dist.all_reduce(out)` of RowParallelLinear.forward (layers/linear.py:228-232) for decode-size
    inputs (T <= t_cap rows).  A skinny matrix-core GEMM whose epilogue pushes each output tile to every rank through
    xGMI peer pointers and adds the ranks' tiles in rank order - every rank ends with bit-identical results.

    Set-up (collective over `group`, once): each rank allocates a receive buffer and a flag array in uncached device
    memory, and the ranks swap inter-process handles with all_gather_object.  `local_world=True` instead builds EVERY
    rank's buffers on this one device (`__call__` then takes lists, one entry per rank, and runs them in one launch):
    the single-GPU parity test, where no second GPU exists."""

    def __init__(self, out_features: int, t_cap: int = 256, group: Optional[dist.ProcessGroup] = None,
                 local_world: int = 0, device=None, device_epoch: bool = True):
        from . import _lib
        self._lib, self._L = _lib, _lib.load()
        self.N, self.t_cap, self.epoch = int(out_features), int(t_cap), 0
        self.local = local_world > 0
        self.world = local_world if self.local else dist.get_world_size(group)
        self.rank = 0 if self.local else dist.get_rank(group)
        if not 1 <= self.world <= _lib.MAX_RANKS:
            raise _lib.MI355AttnError(f"FusedRowParallel: world {self.world} outside [1, {_lib.MAX_RANKS}]")
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        rb = self._L.m355_oproj_recv_bytes(self.world, self.t_cap, self.N)
        fb = self._L.m355_oproj_flag_bytes(self.world, self.t_cap, self.N)
        if rb <= 0 or fb <= 0:
            raise _lib.MI355AttnError("FusedRowParallel: out_features must be a multiple of 64")
        self._own: List[int] = []          # allocations of this process
        self._opened: List[int] = []       # peers' allocations opened here
        self.error = torch.zeros(1, dtype=torch.int32, device=self.device)
        # the calls' epoch lives on the device (a call count per (rank, tile) that the launch advances itself), so a call
        # has the same arguments every time and a captured decode step can replay it
        # (device_epoch=False: the host counts, the other form of the C ABI; every rank must use the same one)
        self.epoch_state = torch.zeros(fb // 4, dtype=torch.int32, device=self.device) if device_epoch else None
        with torch.cuda.device(self.device):
            if self.local:
                self.recv = [self._alloc(rb) for _ in range(self.world)]
                self.flags = [self._alloc(fb) for _ in range(self.world)]
                return
            mine = (self._alloc(rb), self._alloc(fb))
            handles = [None] * self.world
            dist.all_gather_object(handles, tuple(self._export(p) for p in mine), group=group)
            self.recv, self.flags = [0] * self.world, [0] * self.world
            for r, (hr, hf) in enumerate(handles):
                self.recv[r], self.flags[r] = mine if r == self.rank else (self._open(hr), self._open(hf))
            dist.barrier(group=group)

    # Receive buffers and flag arrays are kept for the life of the process: close() hands them to this pool and the next
    # instance of the same size takes them back (zeroed).  Freeing them is what went wrong: on this stack a normal allocation
    # that later lands on the address range of a freed UNCACHED one can be read through stale 4 KiB translations - seen as
    # whole pages of an input tensor holding old data (tests that created and destroyed an instance per case).
    _pool: dict = {}

    def _alloc(self, nbytes: int) -> int:
        spare = FusedRowParallel._pool.get((self.device.index, nbytes))
        if spare:
            ptr = spare.pop()
            with torch.cuda.device(self.device):
                torch.cuda.synchronize()
                self._lib.check(self._L.m355_comm_zero(ptr, nbytes), "m355_comm_zero")
            self._own.append((ptr, nbytes))
            return ptr
        p = ctypes.c_void_p()
        self._lib.check(self._L.m355_comm_alloc(nbytes, ctypes.byref(p)), "m355_comm_alloc")
        self._own.append((p.value, nbytes))
        return p.value

    def _export(self, ptr: int) -> bytes:
        buf = ctypes.create_string_buffer(64)
        self._lib.check(self._L.m355_comm_export(ptr, buf), "m355_comm_export")
        return buf.raw

    _peer_maps: dict = {}             # (device, handle) -> address: a peer's buffer stays mapped once opened (same reason)

    def _open(self, handle: bytes) -> int:
        key = (self.device.index, bytes(handle))
        if key in FusedRowParallel._peer_maps:
            return FusedRowParallel._peer_maps[key]
        p = ctypes.c_void_p()
        self._lib.check(self._L.m355_comm_open(ctypes.create_string_buffer(handle, 64), ctypes.byref(p)), "m355_comm_open")
        FusedRowParallel._peer_maps[key] = p.value
        return p.value

    def close(self) -> None:
        if self._own:
            with torch.cuda.device(self.device):
                torch.cuda.synchronize()
        for p, nbytes in self._own:
            FusedRowParallel._pool.setdefault((self.device.index, nbytes), []).append(p)
        self._opened, self._own = [], []

    def _entries(self, n: int):
        """The call's M355OprojRank array: peers' buffers filled in once, per call only x / w / out change."""
        arr = getattr(self, "_arr", None)
        if arr is None or len(arr) != n:
            arr = self._arr = (self._lib.OprojRank * n)()
            for i in range(n):
                arr[i].rank = i if self.local else self.rank
                for q in range(self.world):
                    arr[i].recv[q], arr[i].flags[q] = self.recv[q], self.flags[q]
        return arr

    def __call__(self, x, w, out=None):
        """x [T, K] and w [N, K] of this rank (lists of them, one per rank, with local_world) -> all-reduced [T, N]."""
        xs, ws = (x, w) if self.local else ((x,), (w,))
        T, K = xs[0].shape
        if T > self.t_cap:
            raise self._lib.MI355AttnError(f"FusedRowParallel: {T} rows > t_cap {self.t_cap} (use row_parallel_linear)")
        if out is None:
            outs = [torch.empty((T, self.N), dtype=xs[0].dtype, device=xs[0].device) for _ in xs]
        else:
            outs = out if self.local else (out,)
        if T == 0:                         # nothing is launched, so this is not a call: the epoch (buffer half) must not move
            return outs if self.local else outs[0]
        arr = self._entries(len(xs))
        for e, xi, wi, oi in zip(arr, xs, ws, outs):
            if not (xi.is_cuda and wi.is_cuda and oi.is_cuda):
                raise self._lib.MI355AttnError("FusedRowParallel: x, w and out must live on the GPU; libmi355attn has no CPU path")
            if xi.shape != (T, K) or wi.shape != (self.N, K) or oi.shape != (T, self.N) or xi.stride(1) != 1 or wi.stride(1) != 1 \
                    or oi.stride(1) != 1 or xi.dtype != wi.dtype or oi.dtype != xi.dtype:
                raise self._lib.MI355AttnError("FusedRowParallel: x [T, K], w [N, K], out [T, N], one dtype, inner stride 1")
            e.x, e.x_row_stride, e.w, e.w_row_stride = xi.data_ptr(), xi.stride(0), wi.data_ptr(), wi.stride(0)
            e.out, e.out_row_stride = oi.data_ptr(), oi.stride(0)
        self.epoch += 1
        self._lib.check(self._L.m355_oproj_allreduce(arr, len(xs), self.world, T, self.N, K, self.t_cap, self.epoch & 0xffffffff,
                                                     None if self.epoch_state is None else self.epoch_state.data_ptr(),
                                                     self.error.data_ptr(),
                                                     self._lib.dtype_code(xs[0].dtype),
                                                     self._lib.current_stream()), "m355_oproj_allreduce")
        return outs if self.local else outs[0]

    def check(self) -> None:
        """Raise if a wait inside any earlier call gave up (a rank never arrived; that call's rows are NaN and this
        instance is out of step with its peers from then on).  Synchronises: call it where the caller syncs anyway."""
        if int(self.error.item()):
            raise self._lib.MI355AttnError("FusedRowParallel: a peer's tile never arrived (bounded wait gave up); the instance "
                                           "is unusable until every rank re-zeroes its buffers")

    def poll_error(self):
        """Non-blocking form for loops that never synchronise (a replayed graph): starts an asynchronous copy of the error
        word and raises if an EARLIER poll's copy has arrived non-zero.  Call every N steps."""
        pend = getattr(self, "_pend", None)
        if pend is not None and pend[1].query():
            if int(pend[0].item()):
                raise self._lib.MI355AttnError("FusedRowParallel: a peer's tile never arrived (bounded wait gave up)")
            pend = None
        if pend is None:
            host = torch.empty(1, dtype=torch.int32, pin_memory=True)
            host.copy_(self.error, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._pend = (host, ev)
