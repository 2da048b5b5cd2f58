"""Req / Batch with the reference's fields and methods (engine/batch.py:20-464), rebuilt so that the
per-batch index metadata costs two host->device copies and ONE kernel launch instead of O(B) tiny tensors,
launches and host syncs.

Two stages:
  * plan_*  - pure host integer work (which rows, which pages, lengths).  No device access; unit-tested on
              CPU against the golden page tables.
  * prepare_for_prefill / prepare_for_decode - upload the plan and run the HIP metadata kernel
              (m355_prefill_prepare / m355_decode_prepare).  GPU only; there is no CPU fallback.

A host mirror of seq_lens (`seq_lens_host`) rides along so that neither the allocator nor the attention
backend ever has to read lengths back from the device (the reference iterates a device tensor element by
element, memory/kvcache.py:44-56, and calls .item() in its metadata, flash_attention_backend.py:60,73).
"""
from __future__ import annotations

import dataclasses
from enum import Enum
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import ops


class Mode(Enum):
    PREFILL = "prefill"
    DECODE = "decode"

    def is_decode(self):
        return self == Mode.DECODE

    def is_prefill(self):
        return self == Mode.PREFILL


@dataclasses.dataclass
class SamplingParams:                      # reference layers/sampler.py:7-12
    max_new_tokens: int = 128
    stop_token_ids: Optional[List[int]] = None
    temperature: float = 0.0
    ignore_eos: bool = False


class FinishReason:
    def __init__(self, kind: str, **info):
        self.kind, self.info, self.is_error = kind, info, kind == "abort"

    def to_json(self):
        return {"type": self.kind, **self.info}


class Req:
    """Request state read by the metadata builders and the radix cache (reference engine/batch.py:91-187)."""

    def __init__(self, rid, origin_input_ids: List[int], sampling_params: Optional[SamplingParams] = None):
        self.rid = rid
        self.origin_input_ids = origin_input_ids
        self.output_ids: List[int] = []
        self.fill_ids: Optional[List[int]] = None
        self.prefix_ppns: Sequence[int] = []
        self.page_table_id: Optional[int] = None
        self.sampling_params = sampling_params or SamplingParams()
        self.last_node = None
        self.finished_reason = None
        self.surr_offset = None
        self.read_offset = None
        self.decoded_text = ""

    def finished(self) -> bool:
        return self.finished_reason is not None

    def adjust_max_prefix_ids(self) -> List[int]:
        """At least one token is always left to extend (batch.py:137-150)."""
        self.fill_ids = self.origin_input_ids + self.output_ids
        return self.fill_ids[: max(len(self.fill_ids) - 1, 0)]

    def init_next_round_input(self, tree_cache) -> None:
        self.fill_ids = self.origin_input_ids + self.output_ids
        self.prefix_ppns, self.last_node = tree_cache.match_prefix(key=self.adjust_max_prefix_ids())

    def check_finished(self) -> None:
        if self.finished():
            return
        if len(self.output_ids) >= self.sampling_params.max_new_tokens:
            self.finished_reason = FinishReason("length", length=self.sampling_params.max_new_tokens)
            return
        stop = self.sampling_params.stop_token_ids
        if stop and self.output_ids[-1] in stop:
            self.finished_reason = FinishReason("stop", matched=self.output_ids[-1])

    def reset_for_retract(self) -> None:
        self.prefix_ppns = []
        self.last_node = None
        self.page_table_id = None


# ------------------------------------------------------------------------------------------------
# host plans
# ------------------------------------------------------------------------------------------------
@dataclasses.dataclass
class PrefillPlan:
    rows: List[int]                 # page-table row per request
    seq_lens: List[int]
    prefix_lens: List[int]
    input_ids: np.ndarray           # int64 [T]
    pages: np.ndarray               # int32: concat over requests of cat(prefix_ppns, new_pages)
    page_off: np.ndarray            # int32 [B+1]
    cu_q: np.ndarray                # int32 [B+1]

    @property
    def T(self) -> int:
        return int(self.cu_q[-1])

    def packed_i32(self) -> np.ndarray:
        """[plan (B+1)x4 | cu_q (B+1) | pages] as one int32 buffer (one H2D copy)."""
        B = len(self.rows)
        plan = np.zeros((B + 1, 4), dtype=np.int32)
        plan[:B, 0] = self.rows
        plan[:B, 1] = self.prefix_lens
        plan[:B, 2] = self.seq_lens
        plan[:, 3] = self.page_off
        return np.concatenate([plan.reshape(-1), self.cu_q, self.pages]).astype(np.int32, copy=False)


def plan_prefill(reqs: List[Req], page_manager, kvcache) -> PrefillPlan:
    """Host half of Batch.prepare_for_prefill (engine/batch.py:244-292): same rows, same pages, same order."""
    ps = page_manager.page_size
    seq_lens = [len(r.fill_ids) for r in reqs]                                       # :255
    prefix_lens = [len(r.prefix_ppns) * ps for r in reqs]                            # :256
    # Everything that can fail is checked BEFORE a row or a page is taken, so a refused batch leaks nothing.
    width = getattr(page_manager, "max_page_num", None)
    if width is not None and seq_lens and -(-max(seq_lens) // ps) > width:
        raise RuntimeError(f"request of {max(seq_lens)} tokens needs more than the page table's {width} columns "
                           "(max_context_len); the reference fails on the row write, memory/page_manager.py:46")
    need = sum(-(-s // ps) - -(-p // ps) for s, p in zip(seq_lens, prefix_lens))
    if need > kvcache.available_page_num():
        raise RuntimeError("KV pool exhausted during prefill (the reference would crash on None pages here)")
    rows = page_manager.allocate(len(reqs))                                          # batch.py:248
    fresh = kvcache.allocate_pages_prefill(seq_lens, prefix_lens)                    # :276-278
    ids, pages, page_off, cu_q = [], [], [0], [0]
    for r, row, pl, sl, new in zip(reqs, rows, prefix_lens, seq_lens, fresh):
        r.page_table_id = row                                                        # :282
        ids.extend(r.fill_ids[pl:])                                                  # :251-253
        pages.extend(int(p) for p in r.prefix_ppns)                                  # :284-292
        pages.extend(new)
        page_off.append(len(pages))
        cu_q.append(cu_q[-1] + sl - pl)
    return PrefillPlan(rows, seq_lens, prefix_lens, np.asarray(ids, dtype=np.int64),
                       np.asarray(pages, dtype=np.int32), np.asarray(page_off, dtype=np.int32),
                       np.asarray(cu_q, dtype=np.int32))


def _check_decode_width(lens: np.ndarray, page_size: int, page_manager) -> None:
    """A decode step must not grow a request past the page table's width (the reference's scatter raises an index
    error there, memory/page_manager.py:54)."""
    width = getattr(page_manager, "max_page_num", None)
    if width is not None and lens.size and -(-int(lens.max()) // page_size) > width:
        raise RuntimeError(f"decode step would grow a request to {int(lens.max())} tokens: more than the page table's "
                           f"{width} columns (max_context_len)")


def plan_decode(seq_lens_host: List[int], kvcache, page_manager=None) -> tuple:
    """Host half of Batch.prepare_for_decode (engine/batch.py:325-337): bump the lengths and pop one page for
    every request whose new token starts a page, in batch order like the reference's per-request loop
    (memory/kvcache.py:49-56: ceil(s/ps) - ceil((s-1)/ps) pages, i.e. one iff (s-1) % ps == 0) - vectorised, no
    per-request Python.  Returns (new seq_lens, int32 [B] new page or -1)."""
    lens = np.asarray(seq_lens_host, dtype=np.int64) + 1
    _check_decode_width(lens, kvcache.page_size, page_manager)
    need = (lens - 1) % kvcache.page_size == 0
    pages = kvcache._allocate_pages(int(need.sum()))
    if pages is None:
        raise RuntimeError("KV pool exhausted during decode (call check_decode_mem / retract_decode first)")
    new_pages = np.full(lens.shape[0], -1, dtype=np.int32)
    new_pages[need] = pages
    return lens.tolist(), new_pages


def _h2d(arr: np.ndarray, device) -> torch.Tensor:
    t = torch.from_numpy(arr)
    if torch.cuda.is_available():
        t = t.pin_memory()
    return t.to(device, non_blocking=True)


_bid = 0


class Batch:
    def __init__(self, reqs: List[Req], page_manager=None, kvcache=None, tree_cache=None, device: str = "cuda"):
        global _bid
        _bid += 1
        self.bid = _bid
        self.reqs = reqs
        self.page_manager = page_manager
        self.page_size = page_manager.page_size if page_manager else None
        self.kvcache = kvcache
        self.tree_cache = tree_cache
        self.device = device
        self.mode: Optional[Mode] = None
        self.page_table_ids = self.seq_lens = self.prefix_lens = None
        self.input_ids = self.output_ids = self.out_cache_loc = self.positions = None
        self.attn_backend = None
        # extras the HIP backend uses (absent on a reference Batch; the backend then derives them)
        self.seq_lens_host: Optional[List[int]] = None
        self.prefix_lens_host: Optional[List[int]] = None
        self.cu_q: Optional[torch.Tensor] = None            # int32 [B+1], prefill only
        self.row_of_token: Optional[torch.Tensor] = None    # int64 [T], prefill only

    def is_empty(self):
        return len(self.reqs) == 0

    def batch_size(self):
        return len(self.reqs)

    # ---- prefill (engine/batch.py:244-317)
    def prepare_for_prefill(self) -> None:
        self.mode = Mode.PREFILL
        plan = plan_prefill(self.reqs, self.page_manager, self.kvcache)
        B, T = len(self.reqs), plan.T
        i64 = _h2d(np.concatenate([plan.input_ids, np.asarray(plan.rows, dtype=np.int64),
                                   np.asarray(plan.seq_lens, dtype=np.int64),
                                   np.asarray(plan.prefix_lens, dtype=np.int64)]), self.device)
        self.input_ids, self.page_table_ids, self.seq_lens, self.prefix_lens = (
            i64[:T], i64[T:T + B], i64[T + B:T + 2 * B], i64[T + 2 * B:T + 3 * B])
        i32 = _h2d(plan.packed_i32(), self.device)
        n_plan = (B + 1) * 4
        d_plan, self.cu_q, d_pages = i32[:n_plan], i32[n_plan:n_plan + B + 1], i32[n_plan + B + 1:]
        self.positions, self.row_of_token, self.out_cache_loc = ops.prefill_prepare(
            self.page_manager.page_table, d_plan, d_pages, self.cu_q, B, T, self.page_size)
        self.seq_lens_host, self.prefix_lens_host = list(plan.seq_lens), list(plan.prefix_lens)

    # ---- decode (engine/batch.py:319-343)
    def prepare_for_decode(self) -> None:
        self.mode = Mode.DECODE
        if self.seq_lens_host is None:
            self.seq_lens_host = [int(x) for x in self.seq_lens.tolist()]
        self.input_ids = self.output_ids
        self.output_ids = None
        if getattr(self.kvcache, "ring", None) is not None:
            self.advance_host_for_decode()
            self.seq_lens, self.positions, self.out_cache_loc = ops.decode_prepare_ring(
                self.page_manager.page_table, self.page_table_ids, self.seq_lens, self.kvcache.ring, self.page_size)
        else:
            self.seq_lens_host, new_pages = plan_decode(self.seq_lens_host, self.kvcache, self.page_manager)
            self.seq_lens, self.positions, self.out_cache_loc = ops.decode_prepare(
                self.page_manager.page_table, self.page_table_ids, self.seq_lens, _h2d(new_pages, self.device),
                self.page_size)
        self.cu_q = self.row_of_token = None

    def advance_host_for_decode(self) -> None:
        """Host half of a decode step with the device free list: count the pages the step takes, move the host mirror
        of the FIFO, bump the mirrored lengths.  (The device half is m355_decode_prepare_ring - eager or replayed
        from a captured graph.)"""
        if self.seq_lens_host is None:
            self.seq_lens_host = [int(x) for x in self.seq_lens.tolist()]
        lens = np.asarray(self.seq_lens_host, dtype=np.int64)
        _check_decode_width(lens + 1, self.page_size, self.page_manager)
        if not self.kvcache.take_on_device(int((lens % self.page_size == 0).sum())):
            raise RuntimeError("KV pool exhausted during decode (call check_decode_mem / retract_decode first)")
        self.seq_lens_host = (lens + 1).tolist()

    # ---- bookkeeping kept working (engine/batch.py:345-464)
    def filter_batch(self, keep_indices: Optional[List[int]] = None) -> None:
        if keep_indices is None:
            keep_indices = [i for i, r in enumerate(self.reqs) if not r.finished()]
        if not keep_indices:
            self.reqs = []
            return
        if len(keep_indices) == len(self.reqs):
            return
        keep = torch.tensor(keep_indices, dtype=torch.int64).to(self.device, non_blocking=True)
        self.reqs = [self.reqs[i] for i in keep_indices]
        self.page_table_ids = self.page_table_ids[keep]
        self.seq_lens = self.seq_lens[keep]
        if self.prefix_lens is not None:
            self.prefix_lens = self.prefix_lens[keep]
        if self.input_ids is not None and self.mode is not None and self.mode.is_decode():
            self.input_ids = self.input_ids[keep]
        if self.output_ids is not None:
            self.output_ids = self.output_ids[keep]
        if self.seq_lens_host is not None:
            self.seq_lens_host = [self.seq_lens_host[i] for i in keep_indices]
        if self.prefix_lens_host is not None:
            self.prefix_lens_host = [self.prefix_lens_host[i] for i in keep_indices]

    def merge_batch(self, other: "Batch") -> None:
        self.page_table_ids = torch.cat([self.page_table_ids, other.page_table_ids])
        self.seq_lens = torch.cat([self.seq_lens, other.seq_lens])
        if self.prefix_lens is not None and other.prefix_lens is not None:
            self.prefix_lens = torch.cat([self.prefix_lens, other.prefix_lens])
        self.out_cache_loc = None
        if self.output_ids is not None:
            self.output_ids = torch.cat([self.output_ids, other.output_ids])
        if self.seq_lens_host is not None and other.seq_lens_host is not None:
            self.seq_lens_host = self.seq_lens_host + other.seq_lens_host
        else:
            self.seq_lens_host = None
        if self.prefix_lens_host is not None and other.prefix_lens_host is not None:
            self.prefix_lens_host = self.prefix_lens_host + other.prefix_lens_host
        self.reqs.extend(other.reqs)

    def _lens_host(self) -> List[int]:
        if self.seq_lens_host is None:
            self.seq_lens_host = [int(x) for x in self.seq_lens.tolist()]
        return self.seq_lens_host

    def check_decode_mem(self, buf_multiplier: float = 1.0) -> bool:
        """Enough free pages for the next decode step? (batch.py:381-391; evicts from the radix tree if one is
        attached - the reference dereferences a None tree_cache here when built without one.)"""
        need = sum(1 for s in self._lens_host() if s % self.page_size == 0) * buf_multiplier
        if self.kvcache.available_page_num() >= need:
            return True
        if self.tree_cache is not None:
            self.tree_cache.evict(need)
        return self.kvcache.available_page_num() >= need

    def retract_decode(self, retract_decode_steps: int = 20):
        """Give the pages of the most recently started requests back until the rest can keep decoding
        (batch.py:393-464, with its undefined `residual_size` resolved to the evident max(0, .) clamp, and the request's
        partly filled last page returned too: the reference's floor at :428-431 leaks it)."""
        order = sorted(range(len(self.reqs)),
                       key=lambda i: (len(self.reqs[i].output_ids), -len(self.reqs[i].origin_input_ids)), reverse=True)
        lens = self._lens_host()
        ps = self.page_size

        def required(n):
            return n * (retract_decode_steps + ps - 1) // ps

        retracted, first = [], True
        while self.kvcache.available_page_num() < required(len(order)) or first:
            if len(order) == 1:
                assert self.kvcache.available_page_num() > 0, "No space left for only one request"
                break
            first = False
            idx = order.pop()
            req = self.reqs[idx]
            retracted.append(req)
            ppns = self.page_manager.page_table[req.page_table_id, len(req.prefix_ppns): -(-lens[idx] // ps)]
            self.kvcache._free_pages(ppns.tolist())
            self.page_manager.free([req.page_table_id])
            if self.tree_cache is not None:
                self.tree_cache.dec_lock_ref(req.last_node)
                residual = max(0, required(len(order)) - self.kvcache.available_page_num())
                self.tree_cache.evict(residual)
            req.reset_for_retract()
        self.filter_batch(sorted(order))          # retracted requests go back to the caller's waiting queue
        done = sum(len(r.output_ids) for r in self.reqs)
        budget = sum(r.sampling_params.max_new_tokens for r in self.reqs)
        ratio = min(1.0, (done + retract_decode_steps * len(self.reqs)) / max(budget, 1))
        return retracted, ratio
