"""Oracle: KV slot write + paged attention, restating the reference's torch-native backend.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  Runs on CPU tensors.

Semantics pinned here (SURVEY.md §8a notes):
  (i)   softmax scale is always ``head_dim ** -0.5`` (``layer.scale`` is never passed on,
        reference torch_native_backend.py:222-234, 264-274);
  (ii)  GQA: q-head ``h`` reads kv-head ``h // (Hq // Hkv)`` (``enable_gqa``, :215, :257);
  (iii) decode attends ALL ``seq_lens[i]`` cached tokens, the just-written one included (:161-186);
  (iv)  extend reads the new tokens' K/V back from the pool, so the slot write comes first (:203-220);
  (v)   packed queries are in batch order (:66, :75, :118);
  (vi)  only page-table entries ``[0, ceil(L/ps))`` of a row are dereferenced (:91-94, :171-174).
"""
from __future__ import annotations

import math

import torch
from torch.nn.functional import scaled_dot_product_attention


def write_kv_ref(k_pool: torch.Tensor, v_pool: torch.Tensor, loc: torch.Tensor,
                 k_new: torch.Tensor, v_new: torch.Tensor) -> None:
    """Row scatter of the new tokens' K/V into the pools (reference memory/kvcache.py:69-79).

    k_pool/v_pool: [S, Hkv, D]; loc: [T] int64 slot ids; k_new/v_new: [T, Hkv, D] (any strides).
    A dtype cast happens only if the source dtype differs from the pool's (index_put semantics).
    """
    k_pool[loc] = k_new
    v_pool[loc] = v_new


def token_slots_ref(page_table: torch.Tensor, row: int, length: int, page_size: int) -> torch.Tensor:
    """Slots of tokens 0..length-1 of one request (reference torch_native_backend.py:91-94, 171-174)."""
    pos = torch.arange(int(length), device=page_table.device)
    return page_table[int(row), pos // page_size] * page_size + (pos % page_size)


def _heads_first(x: torch.Tensor) -> torch.Tensor:
    # [tokens, heads, D] -> [heads, tokens, D]   (reference :64, :96-97, :153)
    return x.movedim(0, x.dim() - 2)


def decode_attn_ref(q: torch.Tensor, k_pool: torch.Tensor, v_pool: torch.Tensor,
                    page_table: torch.Tensor, page_table_ids: torch.Tensor, seq_lens: torch.Tensor,
                    page_size: int, compute_dtype: torch.dtype | None = None) -> torch.Tensor:
    """Decode attention, one sequence at a time (reference torch_native_backend.py:121-193, 237-276).

    q: [B, Hq, D]; returns [B, Hq, D] in q's dtype.  ``compute_dtype=torch.float32`` upcasts
    q/K/V before SDPA (the tolerance anchor); ``None`` computes in the tensors' own dtype like
    the reference does.
    """
    B, Hq, D = q.shape
    Hkv = k_pool.shape[1]
    out = torch.empty_like(q)
    qh = _heads_first(q)
    for i in range(B):
        L = int(seq_lens[i])
        slots = token_slots_ref(page_table, int(page_table_ids[i]), L, page_size)
        k = _heads_first(k_pool[slots])
        v = _heads_first(v_pool[slots])
        qi = qh[:, i:i + 1, :]
        if compute_dtype is not None:
            qi, k, v = qi.to(compute_dtype), k.to(compute_dtype), v.to(compute_dtype)
        o = scaled_dot_product_attention(qi.unsqueeze(0), k.unsqueeze(0), v.unsqueeze(0),
                                         enable_gqa=(Hq != Hkv), scale=None, is_causal=False)
        out[i:i + 1] = o.squeeze(0).movedim(1, 0).to(q.dtype)
    return out


def extend_attn_ref(q: torch.Tensor, k_pool: torch.Tensor, v_pool: torch.Tensor,
                    page_table: torch.Tensor, page_table_ids: torch.Tensor, seq_lens: torch.Tensor,
                    prefix_lens: torch.Tensor, page_size: int,
                    compute_dtype: torch.dtype | None = None) -> torch.Tensor:
    """Ragged prefill over cached prefixes (reference torch_native_backend.py:25-119, 195-235).

    q: [T, Hq, D] with T = sum(seq_lens - prefix_lens), sequences packed in batch order.
    Like the reference, the query block of a sequence is padded at the front to the full KV
    length and run with top-left ``is_causal``; rows ``[prefix, L)`` are kept.  The pad rows are
    zeros here (uninitialised memory in the reference, :79-85) - they are discarded either way.
    """
    T, Hq, D = q.shape
    Hkv = k_pool.shape[1]
    out = torch.empty_like(q)
    qh = _heads_first(q)
    start = 0
    for i in range(seq_lens.shape[0]):
        L = int(seq_lens[i])
        P = int(prefix_lens[i])
        E = L - P
        slots = token_slots_ref(page_table, int(page_table_ids[i]), L, page_size)
        k = _heads_first(k_pool[slots])
        v = _heads_first(v_pool[slots])
        q_pad = torch.zeros((Hq, L, D), dtype=q.dtype, device=q.device)
        q_pad[:, P:, :] = qh[:, start:start + E, :]
        if compute_dtype is not None:
            q_pad, k, v = q_pad.to(compute_dtype), k.to(compute_dtype), v.to(compute_dtype)
        o = scaled_dot_product_attention(q_pad.unsqueeze(0), k.unsqueeze(0), v.unsqueeze(0),
                                         enable_gqa=(Hq != Hkv), scale=None, is_causal=True)
        out[start:start + E] = o.squeeze(0).movedim(1, 0)[P:].to(q.dtype)
        start += E
    return out


def dense_attn_f64(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, q_pos0: int) -> torch.Tensor:
    """Textbook softmax(QK^T/sqrt(D))V in float64 for ONE sequence; the independent anchor.

    q: [E, Hq, D] (query j sits at absolute position q_pos0 + j), k/v: [L, Hkv, D]; query j sees
    keys 0..q_pos0+j.  Decode is the case E == 1, q_pos0 == L-1.
    """
    E, Hq, D = q.shape
    L, Hkv, _ = k.shape
    G = Hq // Hkv
    qd, kd, vd = q.double(), k.double(), v.double()
    kd = kd.repeat_interleave(G, dim=1)
    vd = vd.repeat_interleave(G, dim=1)
    s = torch.einsum("ehd,lhd->hel", qd, kd) / math.sqrt(D)
    key = torch.arange(L).view(1, 1, L)
    qpos = (q_pos0 + torch.arange(E)).view(1, E, 1)
    s = s.masked_fill(key > qpos, float("-inf"))
    p = torch.softmax(s, dim=-1)
    return torch.einsum("hel,lhd->ehd", p, vd)


def paged_attn_f64(q: torch.Tensor, k_pool: torch.Tensor, v_pool: torch.Tensor,
                   page_table: torch.Tensor, page_table_ids: torch.Tensor, seq_lens: torch.Tensor,
                   prefix_lens: torch.Tensor | None, page_size: int) -> torch.Tensor:
    """float64 anchor over a whole batch: decode when ``prefix_lens is None`` else extend."""
    outs = []
    start = 0
    for i in range(seq_lens.shape[0]):
        L = int(seq_lens[i])
        P = L - 1 if prefix_lens is None else int(prefix_lens[i])
        E = L - P
        slots = token_slots_ref(page_table, int(page_table_ids[i]), L, page_size)
        outs.append(dense_attn_f64(q[start:start + E], k_pool[slots], v_pool[slots], P))
        start += E
    return torch.cat(outs, dim=0)


def rope_neox_ref(positions: torch.Tensor, q: torch.Tensor, k: torch.Tensor, head_dim: int,
                  cos_sin_cache: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    """Neox-style rotary embedding with an fp32 cos|sin cache (reference
    layers/rotary_embedding.py:8-18 formula, :45-63 cache layout [max_pos, rotary_dim] = cos|sin).

    q: [T, Hq*D], k: [T, Hkv*D]; rotation computed in float32 and cast back to the input dtype.
    Parity with the third-party ``sgl_kernel`` kernel the reference actually calls (:76-82) is
    UNPINNED: that wheel is not available; this follows the formula the reference keeps in-tree.
    """
    half = head_dim // 2
    cs = cos_sin_cache[positions].float()          # [T, D]
    cos, sin = cs[:, :half].unsqueeze(1), cs[:, half:].unsqueeze(1)

    def rot(x):
        T = x.shape[0]
        xf = x.reshape(T, -1, head_dim).float()
        x1, x2 = xf[..., :half], xf[..., half:]
        o1 = x1 * cos - x2 * sin
        o2 = x2 * cos + x1 * sin
        return torch.cat((o1, o2), dim=-1).to(x.dtype).reshape(x.shape)

    return rot(q), rot(k)


def rope_inv_freq_ref(rotary_dim: int, base: float, rope_scaling: dict | None = None) -> torch.Tensor:
    """Inverse frequencies (reference layers/rotary_embedding.py:45-53) with the Llama-3 rescaling of
    Llama3RotaryEmbedding._compute_inv_freq (:116-137) when ``rope_scaling`` has rope_type 'llama3'."""
    inv = 1.0 / (base ** (torch.arange(0, rotary_dim, 2, dtype=torch.float) / rotary_dim))
    if rope_scaling is None:
        return inv
    factor = rope_scaling["factor"]
    low, high = rope_scaling["low_freq_factor"], rope_scaling["high_freq_factor"]
    orig = rope_scaling["original_max_position_embeddings"]
    low_wl, high_wl = orig / low, orig / high
    wl = 2 * math.pi / inv
    smooth = (orig / wl - low) / (high - low) if low != high else 0
    mid = (1 - smooth) * inv / factor + smooth * inv
    return torch.where(wl < high_wl, inv, torch.where(wl > low_wl, inv / factor, mid))


def rope_cache_ref(rotary_dim: int, max_position: int, base: float, rope_scaling: dict | None = None) -> torch.Tensor:
    """fp32 [max_position, rotary_dim] cos|sin cache (reference layers/rotary_embedding.py:54-63)."""
    inv = rope_inv_freq_ref(rotary_dim, base, rope_scaling)
    t = torch.arange(max_position, dtype=torch.float)
    freqs = torch.einsum("i,j -> ij", t, inv)
    return torch.cat((freqs.cos(), freqs.sin()), dim=-1)
