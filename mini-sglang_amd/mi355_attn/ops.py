"""Tensor-level wrappers over the C ABI (one per entry point of include/mi355attn.h).

Every argument tensor must be on the GPU; results are enqueued on torch's current stream.
"""
from __future__ import annotations

import torch

from . import _lib
from ._lib import check, current_stream, dtype_code, require_cuda


def _ptr(t):
    return None if t is None else t.data_ptr()


def pool_pitch(k_pool: torch.Tensor, v_pool: torch.Tensor) -> int:
    """Row pitch (elements) of a pair of [S, Hkv, D] pools: dense inside a token row, any pitch >= Hkv*D between rows
    (the reference's tensors are contiguous, memory/kvcache.py:24-35; mi355_attn.KVCache pads its rows)."""
    S, Hkv, D = k_pool.shape
    for t in (k_pool, v_pool):
        if t.shape != k_pool.shape or t.stride(2) != 1 or t.stride(1) != D or (S > 1 and t.stride(0) != k_pool.stride(0)):
            raise _lib.MI355AttnError("KV pools must be [S, Hkv, D], dense inside a row, K and V with the same row pitch")
    pitch = k_pool.stride(0) if S > 1 else Hkv * D
    if pitch < Hkv * D or pitch % 8:
        raise _lib.MI355AttnError(f"KV pool row pitch {pitch} must be a multiple of 8 elements and >= Hkv*D")
    return pitch


def kv_store(k_pool: torch.Tensor, v_pool: torch.Tensor, loc: torch.Tensor, k_src: torch.Tensor,
             v_src: torch.Tensor) -> None:
    """k_pool[loc] = k_src; v_pool[loc] = v_src  (reference memory/kvcache.py:69-79).

    k_pool/v_pool [S, Hkv, D] contiguous; k_src/v_src [T, Hkv, D] or [T, Hkv*D], rows may be strided
    (views of the packed qkv projection) but the inner Hkv*D elements must be contiguous."""
    lib = _lib.load()
    for n, t in (("k_pool", k_pool), ("v_pool", v_pool), ("loc", loc), ("k_src", k_src), ("v_src", v_src)):
        require_cuda(t, n)
    S, Hkv, D = k_pool.shape
    T = loc.numel()
    k2, v2 = k_src.view(T, Hkv * D), v_src.view(T, Hkv * D)      # raises if the inner Hkv*D block is not dense
    if k2.dtype != k_pool.dtype or v2.dtype != v_pool.dtype:
        # index_put casts the source to the pool dtype (kvcache.py:78-79); do the same, explicitly
        k2, v2 = k2.to(k_pool.dtype), v2.to(v_pool.dtype)
    if T and (k2.stride(1) != 1 or v2.stride(1) != 1):
        raise _lib.MI355AttnError("kv_store: source rows must have inner stride 1")
    if loc.dtype != torch.int64:
        raise _lib.MI355AttnError("kv_store: loc must be int64")
    if not loc.is_contiguous():
        raise _lib.MI355AttnError("kv_store: loc must be contiguous")
    check(lib.m355_kv_store(k_pool.data_ptr(), v_pool.data_ptr(), pool_pitch(k_pool, v_pool), k2.data_ptr(), v2.data_ptr(),
                            k2.stride(0) if T else Hkv * D, v2.stride(0) if T else Hkv * D, loc.data_ptr(), T, Hkv, D,
                            S, dtype_code(k_pool.dtype), current_stream()), "m355_kv_store")


def out_cache_loc(page_table: torch.Tensor, row_of_token: torch.Tensor, positions: torch.Tensor,
                  page_size: int) -> torch.Tensor:
    """page_table[row, pos // ps] * ps + pos % ps as int64 (reference engine/batch.py:315-317, 341-343)."""
    lib = _lib.load()
    for n, t in (("page_table", page_table), ("row_of_token", row_of_token), ("positions", positions)):
        require_cuda(t, n)
    assert page_table.dtype == torch.int32 and row_of_token.dtype == torch.int64 and positions.dtype == torch.int64
    n = positions.numel()
    out = torch.empty(n, dtype=torch.int64, device=positions.device)
    check(lib.m355_out_cache_loc(page_table.data_ptr(), page_table.stride(0), row_of_token.contiguous().data_ptr(),
                                 positions.contiguous().data_ptr(), page_size, n, out.data_ptr(), current_stream()),
          "m355_out_cache_loc")
    return out


def prefill_prepare(page_table: torch.Tensor, plan: torch.Tensor, pages: torch.Tensor, cu_q: torch.Tensor,
                    B: int, T: int, page_size: int):
    """One launch: page-table row writes + positions + row_of_token + out_cache_loc
    (reference engine/batch.py:280-317).  Returns (positions, row_of_token, out_cache_loc), all int64 [T]."""
    lib = _lib.load()
    for n, t in (("page_table", page_table), ("plan", plan), ("pages", pages), ("cu_q", cu_q)):
        require_cuda(t, n)
    assert plan.dtype == torch.int32 and pages.dtype == torch.int32 and cu_q.dtype == torch.int32
    dev = page_table.device
    positions = torch.empty(T, dtype=torch.int64, device=dev)
    row_of_token = torch.empty(T, dtype=torch.int64, device=dev)
    loc = torch.empty(T, dtype=torch.int64, device=dev)
    check(lib.m355_prefill_prepare(page_table.data_ptr(), page_table.stride(0), page_table.shape[1], plan.data_ptr(), pages.data_ptr(),
                                   cu_q.data_ptr(), B, page_size, positions.data_ptr(), row_of_token.data_ptr(),
                                   loc.data_ptr(), current_stream()), "m355_prefill_prepare")
    return positions, row_of_token, loc


def decode_prepare(page_table: torch.Tensor, rows: torch.Tensor, seq_lens: torch.Tensor, new_pages: torch.Tensor,
                   page_size: int):
    """One launch: seq_lens + 1 (new tensor, like the reference), page-table scatter of fresh pages, positions,
    out_cache_loc (reference engine/batch.py:319-343).  Returns (new_seq_lens, positions, out_cache_loc)."""
    lib = _lib.load()
    for n, t in (("page_table", page_table), ("rows", rows), ("seq_lens", seq_lens), ("new_pages", new_pages)):
        require_cuda(t, n)
    assert rows.dtype == torch.int64 and seq_lens.dtype == torch.int64 and new_pages.dtype == torch.int32
    assert seq_lens.is_contiguous() and rows.is_contiguous()
    B = seq_lens.numel()
    trio = torch.empty((3, B), dtype=torch.int64, device=seq_lens.device)     # new seq_lens | positions | loc
    check(lib.m355_decode_prepare(page_table.data_ptr(), page_table.stride(0), page_table.shape[1], rows.data_ptr(), seq_lens.data_ptr(),
                                  trio[0].data_ptr(), new_pages.data_ptr(), B, page_size, trio[1].data_ptr(),
                                  trio[2].data_ptr(), current_stream()), "m355_decode_prepare")
    return trio[0], trio[1], trio[2]


class PageRing:
    """The free-page FIFO mirrored on the device (m355_page_ring_*): ring [capacity] int32, state {head, count} int64."""

    def __init__(self, capacity: int, device):
        self.capacity = int(capacity)
        self.ring = torch.empty(self.capacity, dtype=torch.int32, device=device)
        self.state = torch.empty(2, dtype=torch.int64, device=device)
        self.reset()

    def reset(self) -> None:
        check(_lib.load().m355_page_ring_init(self.ring.data_ptr(), self.state.data_ptr(), self.capacity,
                                              current_stream()), "m355_page_ring_init")

    def load(self, pages) -> None:
        """Replace the content: `pages` front to back (benchmarks hand pages out in a shuffled order)."""
        t = torch.as_tensor(pages, dtype=torch.int32)
        assert t.numel() <= self.capacity
        self.ring[: t.numel()].copy_(t)
        self.state.copy_(torch.tensor([0, t.numel()], dtype=torch.int64))

    def pop(self, n: int) -> None:
        """The host allocator took n pages off the front (it keeps their ids)."""
        if n:
            check(_lib.load().m355_page_ring_pop(self.state.data_ptr(), self.capacity, int(n), current_stream()),
                  "m355_page_ring_pop")

    def push(self, pages) -> None:
        """Pages given back, appended in this order."""
        t = torch.as_tensor(pages, dtype=torch.int32)
        if t.numel() == 0:
            return
        if t.device != self.ring.device:
            t = (t.pin_memory() if torch.cuda.is_available() else t).to(self.ring.device, non_blocking=True)
        check(_lib.load().m355_page_ring_push(self.ring.data_ptr(), self.state.data_ptr(), self.capacity, t.data_ptr(),
                                              t.numel(), current_stream()), "m355_page_ring_push")

    def snapshot(self):
        """(free page ids front to back) - a device sync; tests and debugging only."""
        head, count = (int(x) for x in self.state.tolist())
        idx = (head + torch.arange(count, device=self.ring.device)) % self.capacity
        return self.ring[idx].tolist()


def decode_prepare_ring(page_table: torch.Tensor, rows: torch.Tensor, seq_lens: torch.Tensor, ring: PageRing,
                        page_size: int, out=None):
    """decode_prepare with the page allocation done by the kernel from the device FIFO (m355_decode_prepare_ring):
    no page ids cross PCIe.  Returns (new_seq_lens, positions, out_cache_loc); `out` = those three as caller-owned
    int64 [B] buffers (new_seq_lens may be `seq_lens` itself: a captured step updates its lengths in place)."""
    lib = _lib.load()
    for n, t in (("page_table", page_table), ("rows", rows), ("seq_lens", seq_lens)):
        require_cuda(t, n)
    assert rows.dtype == torch.int64 and seq_lens.dtype == torch.int64
    assert seq_lens.is_contiguous() and rows.is_contiguous()
    B = seq_lens.numel()
    if out is not None:
        trio = out
        for t in trio:
            require_cuda(t, "out")
            assert t.dtype == torch.int64 and t.numel() == B and t.is_contiguous()
    else:
        trio = torch.empty((3, B), dtype=torch.int64, device=seq_lens.device)     # new seq_lens | positions | loc
    check(lib.m355_decode_prepare_ring(page_table.data_ptr(), page_table.stride(0), page_table.shape[1], rows.data_ptr(), seq_lens.data_ptr(),
                                       trio[0].data_ptr(), ring.ring.data_ptr(), ring.state.data_ptr(), ring.capacity,
                                       B, page_size, trio[1].data_ptr(), trio[2].data_ptr(), current_stream()),
          "m355_decode_prepare_ring")
    return trio[0], trio[1], trio[2]


def decode_workspace_bytes(B: int, Hq: int, Hkv: int, D: int, max_seq_len: int, pool_row_stride: int = 0) -> int:
    return int(_lib.load().m355_decode_workspace_bytes(B, Hq, Hkv, D, max_seq_len, pool_row_stride))


def decode_attn(q: torch.Tensor, k_pool: torch.Tensor, v_pool: torch.Tensor, page_table: torch.Tensor,
                page_table_ids: torch.Tensor, seq_lens: torch.Tensor, page_size: int, num_heads: int,
                max_seq_len: int, scale: float | None = None,
                workspace: torch.Tensor | None = None, out: torch.Tensor | None = None) -> torch.Tensor:
    """Paged decode attention (reference torch_native_backend.py:121-193), self-contained: the call lays its own plan out
    (callers with a per-forward hook use decode_plan + decode_attn_planned and share the plan between layers).
    q [B, Hq*D] (row-strided ok)."""
    lib = _lib.load()
    for n, t in (("q", q), ("k_pool", k_pool), ("v_pool", v_pool), ("page_table", page_table),
                 ("page_table_ids", page_table_ids), ("seq_lens", seq_lens)):
        require_cuda(t, n)
    S, Hkv, D = k_pool.shape
    B = q.shape[0]
    Hq = num_heads
    q2 = q.view(B, Hq * D)
    if q2.stride(1) != 1:
        q2 = q2.contiguous()
    if out is None:
        out = torch.empty((B, Hq * D), dtype=q.dtype, device=q.device)
    if scale is None:
        scale = float(D) ** -0.5          # SDPA default: the reference never passes layer.scale on
    need = decode_workspace_bytes(B, Hq, Hkv, D, max_seq_len, pool_pitch(k_pool, v_pool))
    if workspace is None or workspace.numel() * workspace.element_size() < need or workspace.data_ptr() % 256:
        workspace = torch.empty(need, dtype=torch.uint8, device=q.device)
    ws_bytes = workspace.numel() * workspace.element_size()
    assert page_table.dtype == torch.int32 and page_table_ids.dtype == torch.int64 and seq_lens.dtype == torch.int64
    check(lib.m355_decode_attn(q2.data_ptr(), q2.stride(0), k_pool.data_ptr(), v_pool.data_ptr(), pool_pitch(k_pool, v_pool),
                               page_table.data_ptr(),
                               page_table.stride(0), page_table_ids.data_ptr(), seq_lens.data_ptr(), out.data_ptr(),
                               out.stride(0), B, Hq, Hkv, D, page_size, scale, max_seq_len,
                               _ptr(workspace), ws_bytes, dtype_code(q.dtype), current_stream()), "m355_decode_attn")
    return out


class DecodePlan:
    """Per-forward decode plan (device buffer + the sizes and the pool row pitch it was built for)."""

    def __init__(self, buf, B, Hkv, D, G, capacity, pitch):
        self.buf, self.B, self.Hkv, self.D, self.G, self.capacity, self.pitch = buf, B, Hkv, D, G, capacity, pitch

    def error(self) -> int:
        """The plan's give-up word (M355_PLAN_ERROR_WORD): non-zero when a bounded cross-workgroup wait of the single-launch
        decode kernel ran out in some launch over this plan - that launch's output is not to be trusted.  Reads device
        memory: a sync, so call it where one is paid anyway (tests, the end of a forward, MI355AttnBackend.check())."""
        return int(self.buf[4 * _lib.PLAN_ERROR_WORD: 4 * _lib.PLAN_ERROR_WORD + 4].view(torch.int32).item())


def decode_plan(page_table: torch.Tensor, page_table_ids: torch.Tensor, seq_lens: torch.Tensor, page_size: int,
                num_heads: int, num_kv_heads: int, head_dim: int, token_capacity: int,
                buf: torch.Tensor | None = None, pool_row_stride: int = 0) -> DecodePlan:
    """Build the per-forward decode plan on the device (no host sync).  token_capacity >= sum(seq_lens).
    pool_row_stride: row pitch (elements) of the KV pools the attention calls will read, pool_pitch(k_pool, v_pool);
    0 = contiguous [S, Hkv, D] rows.  The kernel choice - and the plan's format - follow the pool layout, so a plan is
    only valid for pools of that pitch (decode_attn_planned checks)."""
    lib = _lib.load()
    for n, t in (("page_table", page_table), ("page_table_ids", page_table_ids), ("seq_lens", seq_lens)):
        require_cuda(t, n)
    assert page_table.dtype == torch.int32 and page_table_ids.dtype == torch.int64 and seq_lens.dtype == torch.int64
    B = seq_lens.numel()
    G = num_heads // num_kv_heads
    pitch = int(pool_row_stride) or num_kv_heads * head_dim
    need = int(lib.m355_decode_plan_bytes(B, num_kv_heads, head_dim, G, token_capacity, pitch))
    if buf is None or buf.numel() * buf.element_size() < need:
        buf = torch.empty(need, dtype=torch.uint8, device=seq_lens.device)
    check(lib.m355_decode_plan(seq_lens.data_ptr(), page_table_ids.data_ptr(), page_table.data_ptr(),
                               page_table.stride(0), page_table.shape[1], B, page_size, num_kv_heads, head_dim, G, token_capacity,
                               pitch, buf.data_ptr(), buf.numel() * buf.element_size(), current_stream()), "m355_decode_plan")
    return DecodePlan(buf, B, num_kv_heads, head_dim, G, token_capacity, pitch)


def decode_planned_workspace_bytes(B: int, Hq: int, Hkv: int, D: int, token_capacity: int, pool_row_stride: int = 0) -> int:
    return int(_lib.load().m355_decode_planned_workspace_bytes(B, Hq, Hkv, D, token_capacity, pool_row_stride))


STATIC_SCRATCH_BYTES = 8 * 1024 * 1024        # M355_STATIC_SCRATCH_BYTES (include/mi355attn.h)


def decode_attn_planned(q: torch.Tensor, k_pool: torch.Tensor, v_pool: torch.Tensor, plan: DecodePlan,
                        num_heads: int, scale: float | None = None, workspace: torch.Tensor | None = None,
                        out: torch.Tensor | None = None, k_new: torch.Tensor | None = None,
                        v_new: torch.Tensor | None = None, static_scratch: bool = False) -> torch.Tensor:
    """Planned paged decode attention (same result as decode_attn).  With k_new / v_new ([B, Hkv*D] rows of this
    step, inner stride 1) the KV slot write happens in the same launch (m355_decode_store_attn_planned): the
    result and the pools are those of kv_store(...) followed by the attention.
    static_scratch=True (needs k_new / v_new and a caller-owned `out`): m355_decode_store_attn_planned_static, which
    keeps its partials in the library's own device array - nothing is allocated, by the library or by this wrapper."""
    lib = _lib.load()
    for n, t in (("q", q), ("k_pool", k_pool), ("v_pool", v_pool)):
        require_cuda(t, n)
    S, Hkv, D = k_pool.shape
    B, Hq = q.shape[0], num_heads
    if (B, Hkv, D, Hq // Hkv) != (plan.B, plan.Hkv, plan.D, plan.G):
        raise _lib.MI355AttnError("decode_attn_planned: plan was built for a different batch or head shape")
    pitch = pool_pitch(k_pool, v_pool)
    if pitch != plan.pitch:
        raise _lib.MI355AttnError(f"decode_attn_planned: the plan was built for pools of row pitch {plan.pitch}, these have {pitch} "
                                  "(pass pool_row_stride=pool_pitch(k_pool, v_pool) to decode_plan)")
    q2 = q.view(B, Hq * D)
    if q2.stride(1) != 1:
        q2 = q2.contiguous()
    if out is None:
        out = torch.empty((B, Hq * D), dtype=q.dtype, device=q.device)
    if scale is None:
        scale = float(D) ** -0.5
    if workspace is None and not static_scratch:   # (a caller's workspace is size-checked by the library itself)
        workspace = torch.empty(decode_planned_workspace_bytes(B, Hq, Hkv, D, plan.capacity, pitch), dtype=torch.uint8,
                                device=q.device)
    if static_scratch and k_new is None:
        raise _lib.MI355AttnError("decode_attn_planned: static_scratch is the fused-store form (pass k_new / v_new)")
    if k_new is not None:
        require_cuda(k_new, "k_new"), require_cuda(v_new, "v_new")
        k2, v2 = k_new.view(B, Hkv * D), v_new.view(B, Hkv * D)
        if k2.dtype != k_pool.dtype or v2.dtype != v_pool.dtype or k2.stride(1) != 1 or v2.stride(1) != 1:
            raise _lib.MI355AttnError("decode_attn_planned: k_new / v_new must have the pool dtype and inner stride 1")
        if static_scratch:
            check(lib.m355_decode_store_attn_planned_static(q2.data_ptr(), q2.stride(0), k2.data_ptr(), k2.stride(0),
                                                            v2.data_ptr(), v2.stride(0), k_pool.data_ptr(), v_pool.data_ptr(),
                                                            pitch, plan.buf.data_ptr(), out.data_ptr(), out.stride(0), B, Hq, Hkv, D,
                                                            scale, plan.capacity, dtype_code(q.dtype), current_stream()),
                  "m355_decode_store_attn_planned_static")
            return out
        check(lib.m355_decode_store_attn_planned(q2.data_ptr(), q2.stride(0), k2.data_ptr(), k2.stride(0), v2.data_ptr(),
                                                 v2.stride(0), k_pool.data_ptr(), v_pool.data_ptr(), pitch, plan.buf.data_ptr(),
                                                 out.data_ptr(), out.stride(0), B, Hq, Hkv, D, scale, plan.capacity,
                                                 workspace.data_ptr(), workspace.numel() * workspace.element_size(),
                                                 dtype_code(q.dtype), current_stream()), "m355_decode_store_attn_planned")
        return out
    check(lib.m355_decode_attn_planned(q2.data_ptr(), q2.stride(0), k_pool.data_ptr(), v_pool.data_ptr(), pitch,
                                       plan.buf.data_ptr(), out.data_ptr(), out.stride(0), B, Hq, Hkv, D, scale,
                                       plan.capacity, workspace.data_ptr(),
                                       workspace.numel() * workspace.element_size(), dtype_code(q.dtype),
                                       current_stream()), "m355_decode_attn_planned")
    return out


def prefill_workspace_bytes(T: int, Hq: int, D: int, max_seq_len: int) -> int:
    return int(_lib.load().m355_prefill_workspace_bytes(T, Hq, D, max_seq_len))


def prefill_attn(q: torch.Tensor, k_pool: torch.Tensor, v_pool: torch.Tensor, page_table: torch.Tensor,
                 page_table_ids: torch.Tensor, seq_lens: torch.Tensor, prefix_lens: torch.Tensor,
                 cu_q: torch.Tensor, page_size: int, num_heads: int, max_extend_len: int, max_seq_len: int,
                 scale: float | None = None, workspace: torch.Tensor | None = None,
                 out: torch.Tensor | None = None) -> torch.Tensor:
    """Ragged extend attention over cached prefixes (reference torch_native_backend.py:25-119)."""
    lib = _lib.load()
    for n, t in (("q", q), ("k_pool", k_pool), ("v_pool", v_pool), ("page_table", page_table),
                 ("page_table_ids", page_table_ids), ("seq_lens", seq_lens), ("prefix_lens", prefix_lens),
                 ("cu_q", cu_q)):
        require_cuda(t, n)
    S, Hkv, D = k_pool.shape
    T = q.shape[0]
    B = seq_lens.numel()
    Hq = num_heads
    q2 = q.view(T, Hq * D)
    if q2.stride(1) != 1:
        q2 = q2.contiguous()
    if out is None:
        out = torch.empty((T, Hq * D), dtype=q.dtype, device=q.device)
    if scale is None:
        scale = float(D) ** -0.5
    need = prefill_workspace_bytes(T, Hq, D, max_seq_len)
    if need and (workspace is None or workspace.numel() * workspace.element_size() < need):
        workspace = torch.empty(need, dtype=torch.uint8, device=q.device)
    ws_bytes = 0 if workspace is None else workspace.numel() * workspace.element_size()
    assert cu_q.dtype == torch.int32 and prefix_lens.dtype == torch.int64
    check(lib.m355_prefill_attn(q2.data_ptr(), q2.stride(0), k_pool.data_ptr(), v_pool.data_ptr(), pool_pitch(k_pool, v_pool),
                                page_table.data_ptr(), page_table.stride(0), page_table_ids.data_ptr(),
                                seq_lens.data_ptr(), prefix_lens.data_ptr(), cu_q.data_ptr(), out.data_ptr(),
                                out.stride(0), B, T, max_extend_len, max_seq_len, Hq, Hkv, D, page_size, scale,
                                _ptr(workspace), ws_bytes, dtype_code(q.dtype), current_stream()), "m355_prefill_attn")
    return out


def rope_kv_store(positions: torch.Tensor, q: torch.Tensor, k: torch.Tensor, head_dim: int, cos_sin_cache: torch.Tensor,
                  v: torch.Tensor = None, k_pool: torch.Tensor = None, v_pool: torch.Tensor = None,
                  loc: torch.Tensor = None) -> None:
    """Neox rotary embedding on q [T, Hq*D] and k [T, Hkv*D] IN PLACE (layers/rotary_embedding.py:8-18, 65-93) and,
    when pools are given, the slot write of the rotated k and of v in the same launch (memory/kvcache.py:69-79).
    cos_sin_cache: fp32 [max_position, D] (cos | sin)."""
    lib = _lib.load()
    store = k_pool is not None
    ts = [("positions", positions), ("q", q), ("k", k), ("cos_sin_cache", cos_sin_cache)]
    if store:
        ts += [("v", v), ("k_pool", k_pool), ("v_pool", v_pool), ("loc", loc)]
    for n, t in ts:
        require_cuda(t, n)
    T = positions.numel()
    D = head_dim
    if q.dim() != 2 or k.dim() != 2 or (T and (q.stride(1) != 1 or k.stride(1) != 1)):
        raise _lib.MI355AttnError("rope_kv_store: q and k must be [T, heads*D] with inner stride 1")
    Hq, Hkv = q.shape[1] // D, k.shape[1] // D
    if positions.dtype != torch.int64 or cos_sin_cache.dtype != torch.float32 or cos_sin_cache.shape[1] != D \
            or not cos_sin_cache.is_contiguous() or not positions.is_contiguous():
        raise _lib.MI355AttnError("rope_kv_store: positions int64, cos_sin_cache contiguous fp32 [max_position, D]")
    S = 0
    if store:
        S = k_pool.shape[0]
        if v.dim() != 2 or (T and v.stride(1) != 1) or loc.dtype != torch.int64 or not loc.is_contiguous() \
                or k_pool.dtype != k.dtype:
            raise _lib.MI355AttnError("rope_kv_store: v [T, Hkv*D] inner stride 1, loc int64, pools of k's dtype")
    check(lib.m355_rope_kv_store(q.data_ptr(), q.stride(0) if T else Hq * D, k.data_ptr(), k.stride(0) if T else Hkv * D,
                                 _ptr(v) if store else None, (v.stride(0) if T else Hkv * D) if store else 0,
                                 positions.data_ptr(), cos_sin_cache.data_ptr(), cos_sin_cache.shape[0],
                                 _ptr(k_pool), _ptr(v_pool), pool_pitch(k_pool, v_pool) if store else 0,
                                 _ptr(loc) if store else None, T, Hq, Hkv, D, S,
                                 dtype_code(q.dtype), current_stream()), "m355_rope_kv_store")
