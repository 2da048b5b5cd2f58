"""The operator boundary the Llama layers call (reference layers/attention.py:6-36): per-layer constants plus a
dispatch to `batch.attn_backend.forward(q=, k=, v=, layer=, batch=)`."""
from __future__ import annotations

import torch.nn as nn


class Attention(nn.Module):
    def __init__(self, layer_id: int, num_heads: int, num_kv_heads: int, head_dim: int, scale: float = 1.0):
        super().__init__()
        self.layer_id = layer_id
        self.num_heads = num_heads
        self.num_kv_heads = num_kv_heads
        self.head_dim = head_dim
        self.scale = scale          # kept for interface parity; the reference never forwards it to SDPA

    def forward(self, q, k, v, batch):
        return batch.attn_backend.forward(q=q, k=k, v=v, layer=self, batch=batch)
