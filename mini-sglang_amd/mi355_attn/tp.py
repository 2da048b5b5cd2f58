"""Tensor-parallel head sharding of the attention path.

The reference shards attention by heads with no communication inside the backend: rank r of tp owns the
contiguous q-heads [r*Hq/tp, (r+1)*Hq/tp) and kv-heads [r*Hkv/tp, (r+1)*Hkv/tp) (models/llama.py:76-87,
layers/linear.py:171-190), keeps a KV pool holding only its kv-heads (engine/model_runner.py:161) and an
identical page table on every rank.  The one exchange step is the SUM all-reduce RowParallelLinear issues on
the o_proj output (layers/linear.py:228-232) - RCCL over xGMI on the GPUs ("nccl" backend), gloo in CPU tests.
"""
from __future__ import annotations

import dataclasses
from typing import Optional

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
