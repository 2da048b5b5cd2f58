"""A Llama forward pass with random weights around the MI355X attention path.

Mirrors the call structure of the reference model (models/llama.py:127-138 attention block, :189-209 decoder
layer, :237-250 model, :288-296 causal LM) and of ModelRunner.forward (engine/model_runner.py:189-207): per layer
qkv GEMM -> split -> rotary -> attention backend -> o_proj (+ all-reduce under TP) -> MLP, greedy sampling at
temperature 0.  There are no checkpoints in this environment, so weights are seeded random tensors of the
architecture's shapes; under TP every rank draws the same full tensors and keeps its slice
(column-parallel qkv / gate_up, row-parallel o_proj / down, layers/linear.py:171-232).

The path pieces are this package's HIP kernels: m355_rope_kv_store (rotary + KV slot write in one launch),
MI355AttnBackend (prefill / decode attention), Batch.prepare_for_* (metadata kernels).
"""
from __future__ import annotations

import dataclasses
from typing import Any, Dict, Optional

import torch
import torch.nn.functional as F

import mi355_attn as m
from mi355_attn.rotary import RotaryEmbedding
from mi355_attn.tp import FusedRowParallel, all_reduce_rows, shard_heads


@dataclasses.dataclass(frozen=True)
class ModelShape:
    name: str
    layers: int
    hidden: int
    intermediate: int
    heads: int
    kv_heads: int
    head_dim: int
    vocab: int
    rope_theta: float = 500000.0
    rope_scaling: Optional[Dict[str, Any]] = None
    max_position: int = 8192
    rms_eps: float = 1e-5
    tie_embeddings: bool = False


_LLAMA3_SCALING = {"rope_type": "llama3", "factor": 32.0, "low_freq_factor": 1.0, "high_freq_factor": 4.0,
                   "original_max_position_embeddings": 8192}
SHAPES = {
    # HF public configs
    "llama-3.2-1b": ModelShape("Llama-3.2-1B", 16, 2048, 8192, 32, 8, 64, 128256, rope_scaling=_LLAMA3_SCALING,
                               max_position=131072, tie_embeddings=True),
    "llama-3-8b": ModelShape("Llama-3-8B", 32, 4096, 14336, 32, 8, 128, 128256),
    "llama-3-70b": ModelShape("Llama-3-70B", 80, 8192, 28672, 64, 8, 128, 128256),
    "tiny": ModelShape("tiny", 2, 256, 512, 4, 2, 64, 512, max_position=2048),
    "tiny128": ModelShape("tiny128", 2, 512, 1024, 4, 1, 128, 512, rope_scaling=_LLAMA3_SCALING, max_position=16384),
}


class SyntheticLlama:
    def __init__(self, shape: ModelShape, tp_rank: int = 0, tp_size: int = 1, device: str = "cuda",
                 dtype: torch.dtype = torch.bfloat16, seed: int = 0, max_position: Optional[int] = None,
                 fused_row_parallel: bool = False):
        self.shape, self.device, self.dtype = shape, device, dtype
        # o_proj / down_proj + their all-reduce as one launch for decode-size inputs (TP only)
        self.fused = FusedRowParallel(shape.hidden, t_cap=256, device=device) if fused_row_parallel and tp_size > 1 else None
        self.tp_rank, self.tp_size = tp_rank, tp_size
        self.heads = shard_heads(shape.heads, shape.kv_heads, shape.head_dim, tp_size, tp_rank)
        D, H = shape.head_dim, shape.hidden
        self.q_size, self.kv_size = self.heads.num_heads * D, self.heads.num_kv_heads * D
        if shape.intermediate % tp_size:
            raise ValueError("intermediate size does not divide over tp_size")
        I = shape.intermediate // tp_size
        g = torch.Generator(device=device).manual_seed(seed)

        def w(rows, cols, std):
            return (torch.randn((rows, cols), generator=g, device=device, dtype=torch.float32) * std).to(dtype)

        self.embed = w(shape.vocab, H, 1.0)
        self.lm_head = self.embed if shape.tie_embeddings else w(shape.vocab, H, H ** -0.5)
        self.layers = []
        qc, kc = self.heads.q_columns(), self.heads.kv_columns()
        for _ in range(shape.layers):
            wq, wk, wv = w(shape.heads * D, H, H ** -0.5), w(shape.kv_heads * D, H, H ** -0.5), w(shape.kv_heads * D, H, H ** -0.5)
            wo = w(H, shape.heads * D, (shape.heads * D) ** -0.5)
            wg, wu = w(shape.intermediate, H, H ** -0.5), w(shape.intermediate, H, H ** -0.5)
            wd = w(H, shape.intermediate, shape.intermediate ** -0.5)
            lo, hi = tp_rank * I, (tp_rank + 1) * I
            self.layers.append(dict(
                wqkv=torch.cat([wq[qc], wk[kc], wv[kc]]).contiguous(), wo=wo[:, qc].contiguous(),
                w_gate_up=torch.cat([wg[lo:hi], wu[lo:hi]]).contiguous(), w_down=wd[:, lo:hi].contiguous(),
                ln1=torch.ones(H, device=device, dtype=dtype), ln2=torch.ones(H, device=device, dtype=dtype)))
            del wq, wk, wv, wo, wg, wu, wd
        self.norm = torch.ones(H, device=device, dtype=dtype)
        self.rope = RotaryEmbedding(D, D, max_position or shape.max_position, shape.rope_theta, shape.rope_scaling, device)
        self.attn = [m.Attention(i, self.heads.num_heads, self.heads.num_kv_heads, D) for i in range(shape.layers)]

    def _norm(self, x, residual, weight):
        # RMSNorm with fused residual add (layers/layernorm.py:14-41)
        if residual is not None:
            x = x + residual
        return F.rms_norm(x, (x.shape[-1],), weight, self.shape.rms_eps), x

    def _row_parallel(self, x, w):
        """RowParallelLinear.forward (layers/linear.py:228-232)."""
        # the one-launch GEMM + reduce pays off where the collective's latency dominates: decode-size row counts and a
        # short K per rank (its GEMM streams (16 + 64) x K per CU: 13.6 us at K = 2 048 against 8.2 for hipBLASLt, world 1)
        if self.fused is not None and x.shape[0] <= min(64, self.fused.t_cap) and x.shape[1] <= 2048:
            return self.fused(x if x.stride(-1) == 1 else x.contiguous(), w)
        return all_reduce_rows(F.linear(x, w))

    @torch.no_grad()
    def forward(self, batch) -> torch.Tensor:
        """Hidden states of every token -> logits of each sequence's LAST token, [B, vocab]."""
        x = F.embedding(batch.input_ids, self.embed)
        residual = None
        pos, loc = batch.positions, batch.out_cache_loc
        for lw, attn in zip(self.layers, self.attn):
            h, residual = self._norm(x, residual, lw["ln1"])
            qkv = F.linear(h, lw["wqkv"])
            q, k, v = qkv.split([self.q_size, self.kv_size, self.kv_size], dim=-1)       # models/llama.py:133-134
            k_pool, v_pool = batch.kvcache.get_kv_cache(attn.layer_id)
            self.rope.rotate_and_store(pos, q, k, v, k_pool, v_pool, loc)                 # :135 + kvcache.py:69-79
            o = batch.attn_backend.forward(q=q, k=None, v=None, layer=attn, batch=batch)  # layers/attention.py:30-36
            x = self._row_parallel(o, lw["wo"])                                            # layers/linear.py:228-232
            h, residual = self._norm(x, residual, lw["ln2"])
            gu = F.linear(h, lw["w_gate_up"])
            gate, up = gu.chunk(2, dim=-1)
            x = self._row_parallel(F.silu(gate) * up, lw["w_down"])
        h, _ = self._norm(x, residual, self.norm)
        if batch.mode.is_prefill():
            h = h[(batch.cu_q[1:] - 1).long()]            # last new token of each sequence
        return F.linear(h, self.lm_head)


class HarnessRunner:
    """What bench_one_batch needs of ModelRunner (engine/model_runner.py:53-207): memory pools, the attention
    backend, and forward(batch) -> (logits, next_token_ids)."""

    def __init__(self, shape: ModelShape, page_size: int, max_total_tokens: int, max_reqs: int, context_len: int,
                 tp_rank: int = 0, tp_size: int = 1, device: str = "cuda", dtype: torch.dtype = torch.bfloat16, seed: int = 0,
                 fused_row_parallel: bool = False):
        self.device, self.page_size, self.dtype = device, page_size, dtype
        self.tp_rank, self.tp_size = tp_rank, tp_size
        self.model = SyntheticLlama(shape, tp_rank, tp_size, device, dtype, seed, max_position=max(context_len + 8, 64),
                                    fused_row_parallel=fused_row_parallel)
        size = max_total_tokens // page_size * page_size                                    # model_runner.py:141-145
        self.kvcache = m.KVCache(size=size, page_size=page_size, head_num=self.model.heads.num_kv_heads,
                                 head_dim=shape.head_dim, dtype=dtype, layer_num=shape.layers, device=device,
                                 device_free_list=str(device).startswith("cuda"))
        self.page_manager = m.PageManager(page_size=page_size, max_req_num=max_reqs, max_context_len=context_len + 4,
                                          device=device)
        self.attn_backend = m.MI355AttnBackend(self)                                        # model_runner.py:90-91

    @torch.no_grad()
    def forward(self, batch):
        self.attn_backend.init_forward_metadata(batch)                                      # model_runner.py:194-195
        batch.attn_backend = self.attn_backend
        logits = self.model.forward(batch)
        fused = getattr(self.model, "fused", None)
        if fused is not None and not torch.cuda.is_current_stream_capturing():
            # the fused row-parallel launches' give-up word: an asynchronous copy is started here and the PREVIOUS forward's is
            # examined - no sync; a peer that never arrived (that forward's rows are NaN) raises within a step or two
            fused.poll_error()
        return logits, torch.argmax(logits, dim=-1)                                         # temperature 0 (:201-205)
