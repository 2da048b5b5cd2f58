"""MI355AttnBackend: the attention backend of Mini-SGLang rebuilt on hand-written HIP kernels.

Drop-in for the object constructed at reference engine/model_runner.py:90-91.  Same interface as the two
reference backends (layers/attention_backends/torch_native_backend.py:15-276 and
flash_attention_backend.py:44-151):

    backend = MI355AttnBackend(model_runner)
    backend.init_forward_metadata(batch)                  # once per forward
    o = backend.forward(q, k, v, layer, batch)            # once per layer; also forward_extend / forward_decode

It works with this package's Batch (host mirrors of the lengths ride along, so no device sync happens) and
with the reference's own Batch (then lengths are read back once per forward, as its FA3 backend does).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch

from . import ops


@dataclass
class ForwardMetadata:
    """Built once per forward (the counterpart of FlashAttentionMetaData, flash_attention_backend.py:22-41)."""
    decode: bool
    max_seq_len: int
    max_extend_len: int = 1
    plan: Optional[object] = None                  # ops.DecodePlan, decode only
    cu_q: Optional[torch.Tensor] = None            # int32 [B+1], prefill only
    workspace: Optional[torch.Tensor] = None
    ws: Optional[torch.Tensor] = None              # planned decode: the workspace of this forward ...
    ws_key: Optional[tuple] = None                 # ... and the (B, Hq, Hkv, D) it was sized for
    first_layer: Optional[int] = None              # layer_id of the first layer that used this metadata


_NATIVE_GROUPS = (1, 2, 4, 8)          # query heads per kv head the kernels are built for


def _group_pad(num_heads: int, num_kv_heads: int) -> int:
    """The group size the kernels run a layer at.  The reference takes any Llama config (utils/model_config.py:24-47); the
    HIP kernels are built for groups of 1, 2, 4 and 8 query heads per kv head, so any other group (Llama-3.2-3B: 24 / 8 = 3;
    Yi-34B: 56 / 8 = 7) runs in the next one up with all-zero query heads appended to every group - their outputs are
    dropped - at (padded / real) times the attention arithmetic and two small copies per layer."""
    if num_heads % num_kv_heads:
        raise ops._lib.MI355AttnError(f"MI355AttnBackend: {num_heads} query heads are not a multiple of {num_kv_heads} kv heads")
    g = num_heads // num_kv_heads
    for n in _NATIVE_GROUPS:
        if n >= g:
            return n
    raise ops._lib.MI355AttnError(f"MI355AttnBackend: {g} query heads per kv head (at most {_NATIVE_GROUPS[-1]} are supported)")


def _pad_groups(q: torch.Tensor, Hkv: int, G: int, Gp: int, D: int) -> torch.Tensor:
    T = q.shape[0]
    return torch.nn.functional.pad(q.reshape(T, Hkv, G, D), (0, 0, 0, Gp - G)).view(T, Hkv * Gp * D)


def _drop_pad(o: torch.Tensor, Hkv: int, G: int, Gp: int, D: int) -> torch.Tensor:
    T = o.shape[0]
    return o.view(T, Hkv, Gp, D)[:, :, :G].reshape(T, Hkv * G * D)


class MI355AttnBackend:
    def __init__(self, model_runner):
        self.device = model_runner.device
        self.page_size = model_runner.page_size
        self.forward_metadata: Optional[ForwardMetadata] = None
        self._workspace: Optional[torch.Tensor] = None
        self._plan_buf: Optional[torch.Tensor] = None
        # set while a captured HIP graph holds raw pointers into _workspace / _plan_buf (bench_one_batch.GraphedDecode):
        # replacing either buffer would leave the graph replaying against freed memory, so growth is refused instead
        self.buffers_pinned = False
        self._plan_key = None
        # decode: one plan per forward shared by all layers (m355_decode_plan) + one attention launch per layer that also
        # writes the step's K/V rows (m355_decode_store_attn_planned).  fused_store = False keeps the separate
        # m355_kv_store launch (cross-check tests).
        self.decode_mode = "planned"
        self.fused_store = True
        # planned decode: lay the plan out for at least this many tokens (a captured decode step needs every kernel
        # argument - the plan's layout among them - to stay the same while the batch grows)
        self.decode_capacity_hint = 0

    # ---- metadata -------------------------------------------------------------------------------
    def _ws(self, nbytes: int, device) -> Optional[torch.Tensor]:
        if nbytes == 0:
            return None
        if self._workspace is None or self._workspace.numel() < nbytes:
            if self.buffers_pinned:
                raise ops._lib.MI355AttnError("MI355AttnBackend: a captured graph pins the workspace; this forward needs a "
                                              "bigger one (release the graph first: GraphedDecode.release())")
            self._workspace = torch.empty(int(nbytes * 1.25), dtype=torch.uint8, device=device)
        return self._workspace

    def init_forward_metadata(self, batch) -> None:
        lens = getattr(batch, "seq_lens_host", None)
        if lens is None:
            lens = [int(x) for x in batch.seq_lens.tolist()]          # one sync, like flash_attention_backend.py:60
        if batch.mode.is_decode():
            self.forward_metadata = ForwardMetadata(decode=True, max_seq_len=max(lens))
            self._plan_key = (batch, sum(lens))       # the plan itself needs the layer's head shape: built lazily
            return
        pre = getattr(batch, "prefix_lens_host", None)
        if pre is None:
            pre = [int(x) for x in batch.prefix_lens.tolist()]
        cu_q = getattr(batch, "cu_q", None)
        if cu_q is None:
            ext = (batch.seq_lens - batch.prefix_lens).to(torch.int32)
            cu_q = torch.nn.functional.pad(torch.cumsum(ext, 0, dtype=torch.int32), (1, 0))
        self.forward_metadata = ForwardMetadata(decode=False, max_seq_len=max(lens),
                                                max_extend_len=max(s - p for s, p in zip(lens, pre)), cu_q=cu_q)

    def check(self) -> None:
        """Raise if a launch over the current decode plan gave up a bounded cross-workgroup wait (the plan's
        M355_PLAN_ERROR_WORD; include/mi355attn.h) - its attention output is then wrong.  One 4-byte read-back, i.e. a
        device sync: call it where the engine synchronises anyway (after sampling, scheduler.py:416 `.tolist()`)."""
        md = self.forward_metadata
        if md is not None and md.plan is not None and md.plan.error():
            raise ops._lib.MI355AttnError("MI355AttnBackend: a decode launch gave up waiting for another workgroup's partial "
                                          "result (plan error word set): the outputs of this forward are invalid")

    # ---- per layer ------------------------------------------------------------------------------
    def forward(self, q, k, v, layer, batch):
        if batch.mode.is_decode():
            return self.forward_decode(q, k, v, layer, batch)
        return self.forward_extend(q, k, v, layer, batch)

    def _store_kv(self, k, v, layer, batch):
        if k is None:
            return                                                     # torch_native_backend.py:203, :245
        assert v is not None
        k_pool, v_pool = batch.kvcache.get_kv_cache(layer.layer_id)
        ops.kv_store(k_pool, v_pool, batch.out_cache_loc, k, v)        # replaces KVCache.write_kv

    def _new_forward(self, md, layer) -> bool:
        """True when `layer` is the layer that first used `md`: the model has come round again, i.e. a new forward
        began without init_forward_metadata (legal against the torch-native backend, whose hook is a no-op,
        torch_native_backend.py:21-23).  The per-forward plan / lengths are then stale and must be rebuilt."""
        if md.first_layer is None:
            md.first_layer = layer.layer_id
            return False
        return md.first_layer == layer.layer_id

    def forward_decode(self, q, k, v, layer, batch):
        md = self.forward_metadata
        if md is None or not md.decode or self._new_forward(md, layer):
            self.init_forward_metadata(batch)
            md = self.forward_metadata
            md.first_layer = layer.layer_id
        k_pool, v_pool = batch.kvcache.get_kv_cache(layer.layer_id)
        B = q.shape[0]
        fuse = (self.fused_store and k is not None and k.dtype == k_pool.dtype
                and v.dtype == v_pool.dtype and k.stride(-1) == 1 and v.stride(-1) == 1)
        if not fuse:
            self._store_kv(k, v, layer, batch)
        Hq, Hkv, D = layer.num_heads, layer.num_kv_heads, layer.head_dim
        G, Gp = Hq // Hkv, _group_pad(Hq, Hkv)
        if Gp != G:
            q, Hq = _pad_groups(q.reshape(B, Hq * D), Hkv, G, Gp, D), Hkv * Gp
        if md.plan is None:                        # first layer of this forward: lay the token stream out once
            cap = max(self._plan_key[1], int(self.decode_capacity_hint))
            md.plan = ops.decode_plan(batch.page_manager.page_table, batch.page_table_ids, batch.seq_lens,
                                      self.page_size, Hq, Hkv, D, cap, buf=self._plan_buf,
                                      pool_row_stride=ops.pool_pitch(k_pool, v_pool))
            if self.buffers_pinned and md.plan.buf is not self._plan_buf:
                raise ops._lib.MI355AttnError("MI355AttnBackend: a captured graph pins the plan buffer; this forward needs a "
                                              "bigger one (release the graph first: GraphedDecode.release())")
            self._plan_buf = md.plan.buf
        ws = md.ws                                 # sized once per forward (same shape in every layer)
        if ws is None or md.ws_key != (B, Hq, Hkv, D):
            ws = md.ws = self._ws(max(256, ops.decode_planned_workspace_bytes(B, Hq, Hkv, D, md.plan.capacity, md.plan.pitch)), q.device)
            md.ws_key = (B, Hq, Hkv, D)
        o = ops.decode_attn_planned(q.reshape(B, Hq * D), k_pool, v_pool, md.plan, Hq, workspace=ws,
                                    k_new=k if fuse else None, v_new=v if fuse else None)
        return o if Gp == G else _drop_pad(o, Hkv, G, Gp, D)

    def forward_extend(self, q, k, v, layer, batch):
        md = self.forward_metadata
        if md is None or md.decode or self._new_forward(md, layer):
            self.init_forward_metadata(batch)
            md = self.forward_metadata
            md.first_layer = layer.layer_id
        self._store_kv(k, v, layer, batch)       # first: the new tokens' K/V are read back from the pool
        k_pool, v_pool = batch.kvcache.get_kv_cache(layer.layer_id)
        T = q.shape[0]
        Hq, Hkv, D = layer.num_heads, layer.num_kv_heads, layer.head_dim
        G, Gp = Hq // Hkv, _group_pad(Hq, Hkv)
        if Gp != G:
            q, Hq = _pad_groups(q, Hkv, G, Gp, D), Hkv * Gp
        ws = self._ws(ops.prefill_workspace_bytes(T, Hq, D, md.max_seq_len), q.device)
        o = ops.prefill_attn(q, k_pool, v_pool, batch.page_manager.page_table, batch.page_table_ids, batch.seq_lens,
                             batch.prefix_lens, md.cu_q, self.page_size, Hq, md.max_extend_len, md.max_seq_len, workspace=ws)
        return o if Gp == G else _drop_pad(o, Hkv, G, Gp, D)
