"""Host side of the rotary embedding (reference layers/rotary_embedding.py:21-188): the fp32 cos|sin cache with
optional Llama-3 frequency scaling, and a module-like object whose call rotates q, k in place through the HIP
kernel (m355_rope_kv_store).  The cache is built once at model construction with plain torch math on the CPU and
moved to the GPU; nothing here runs per token."""
from __future__ import annotations

import math
from typing import Any, Dict, Optional

import torch

from . import ops


def compute_inv_freq(rotary_dim: int, base: float, rope_scaling: Optional[Dict[str, Any]] = None) -> torch.Tensor:
    """inv_freq[i] = base^(-2i/rotary_dim) (rotary_embedding.py:45-53), with the llama3 rescaling of
    Llama3RotaryEmbedding._compute_inv_freq (:116-137) when rope_scaling says so."""
    inv = 1.0 / (base ** (torch.arange(0, rotary_dim, 2, dtype=torch.float) / rotary_dim))
    if rope_scaling is None:
        return inv
    kind = rope_scaling.get("rope_type", rope_scaling.get("type"))
    if kind != "llama3":
        raise ValueError(f"Unsupported RoPE scaling type {kind}")                     # :186
    factor, lo, hi = rope_scaling["factor"], rope_scaling["low_freq_factor"], rope_scaling["high_freq_factor"]
    orig = rope_scaling["original_max_position_embeddings"]
    wave_len = 2 * math.pi / inv
    smooth = (orig / wave_len - lo) / (hi - lo) if lo != hi else 0
    return torch.where(wave_len < orig / hi, inv,
                       torch.where(wave_len > orig / lo, inv / factor, (1 - smooth) * inv / factor + smooth * inv))


def compute_cos_sin_cache(rotary_dim: int, max_position: int, base: float,
                          rope_scaling: Optional[Dict[str, Any]] = None) -> torch.Tensor:
    """fp32 [max_position, rotary_dim] = cos(t * inv_freq) | sin(t * inv_freq)  (rotary_embedding.py:54-63)."""
    inv = compute_inv_freq(rotary_dim, base, rope_scaling)
    t = torch.arange(max_position, dtype=torch.float)
    freqs = torch.einsum("i,j -> ij", t, inv)
    return torch.cat((freqs.cos(), freqs.sin()), dim=-1)


class RotaryEmbedding:
    """get_rope(...) counterpart (rotary_embedding.py:139-188) for full-head rotation (rotary_dim == head_size, the
    only case Llama uses).  __call__(positions, q, k) rotates in place and returns (q, k) like the reference;
    rotate_and_store additionally scatters rotated k and v into the KV pools in the same launch."""

    def __init__(self, head_size: int, rotary_dim: int, max_position: int, base: float,
                 rope_scaling: Optional[Dict[str, Any]] = None, device: str = "cuda"):
        if rotary_dim != head_size:
            raise ValueError("partial rotary embeddings are not supported by the HIP kernel")
        self.head_size = head_size
        self.cos_sin_cache = compute_cos_sin_cache(rotary_dim, max_position, base, rope_scaling).to(device)

    def __call__(self, positions: torch.Tensor, query: torch.Tensor, key: torch.Tensor):
        ops.rope_kv_store(positions, query, key, self.head_size, self.cos_sin_cache)
        return query, key

    def rotate_and_store(self, positions, query, key, value, k_pool, v_pool, loc):
        ops.rope_kv_store(positions, query, key, self.head_size, self.cos_sin_cache, value, k_pool, v_pool, loc)
        return query, key
