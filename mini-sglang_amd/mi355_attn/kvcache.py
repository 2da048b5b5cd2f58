"""KV pools + free-page FIFO with the reference's interface (memory/kvcache.py:5-91).

Differences that do not change results: all layers' K and V pools are carved from ONE HBM allocation
([layer, K|V, S, row pitch]; `k_cache[l]` / `v_cache[l]` are views with the reference's [S, Hkv, D] shape), the
free list is a deque (O(1) pops instead of list slicing), and `write_kv` is the HIP row scatter instead of
two `index_put_` launches.

Row pitch: a token row of the reference's contiguous [S, Hkv, D] pool is Hkv*D*2 bytes - a power of two for every
Llama shape - so one kv head's K (or V) bytes of ALL tokens sit in one fixed 128/256-byte column: fixed address bits
7..9.  The decode kernel gives a CU one kv head to stream, and on MI355X a CU whose every request hits one such
column runs up to 30 % slower on some columns (tools/stamp_share.py).  With `row_pad` elements of
padding per row (default: 128 bytes when a row is >= 1 KiB, i.e. <= 12.5 % of the pool) the pitch is an odd number of
128-byte lines and every head's column walks through all eight patterns as the slot number grows.  The views keep the
reference's shape and indexing; only `stride(0)` differs, and every entry point of the library takes that pitch
(`pool_row_stride`).  `row_pad=0` gives the reference's contiguous layout.
"""
from __future__ import annotations

from collections import deque
from typing import Iterable, List, Optional, Sequence, Tuple

import torch

from . import ops


class KVCache:
    def __init__(self, size: int, page_size: int, head_num: int, head_dim: int, dtype: torch.dtype,
                 layer_num: int, device: str, device_free_list: bool = False, row_pad: Optional[int] = None):
        self.page_num = size // page_size                                   # kvcache.py:16
        self.page_size = page_size
        self.head_num = head_num
        self.head_dim = head_dim
        self.dtype = dtype
        self.layer_num = layer_num
        self.device = device
        rows = self.page_num * page_size
        width = head_num * head_dim
        esize = torch.empty((), dtype=dtype).element_size()
        if row_pad is None:                                                 # 128 bytes per row where that costs <= 12.5 %
            row_pad = 128 // esize if width * esize >= 1024 and (width * esize) % 256 == 0 else 0
        if row_pad < 0 or (row_pad * esize) % 16:
            raise ValueError("row_pad must be a whole number of 16-byte chunks")
        self.row_pitch = width + row_pad                                    # elements between consecutive token rows
        self._store = torch.zeros((layer_num, 2, rows, self.row_pitch), dtype=dtype, device=device)
        # the reference's [S, Hkv, D] tensors (kvcache.py:24-35), as views into the padded rows
        self.k_cache = [self._store[l, 0, :, :width].unflatten(1, (head_num, head_dim)) for l in range(layer_num)]
        self.v_cache = [self._store[l, 1, :, :width].unflatten(1, (head_num, head_dim)) for l in range(layer_num)]
        # optional device mirror of the FIFO: decode steps then allocate inside their metadata kernel
        self.ring = ops.PageRing(self.page_num, device) if device_free_list else None
        self._ring_defer = None                    # pops batched into one launch while a prefill allocates
        self.clear()

    # ---- free-page FIFO: pop from the front, return to the back (kvcache.py:38-67, 87-91)
    @property
    def free_pages(self) -> List[int]:
        return list(self._free)

    def allocate_pages_prefill(self, seq_lens: Sequence[int], prefix_lens: Sequence[int]) -> List[Optional[List[int]]]:
        ps = self.page_size
        self._ring_defer = 0
        try:
            return [self._allocate_pages(-(-int(s) // ps) - -(-int(p) // ps)) for s, p in zip(seq_lens, prefix_lens)]
        finally:
            n, self._ring_defer = self._ring_defer, None
            if self.ring is not None:
                self.ring.pop(n)

    def allocate_pages_decode(self, seq_lens: Sequence[int]) -> List[Optional[List[int]]]:
        return self.allocate_pages_prefill(seq_lens, [int(s) - 1 for s in seq_lens])

    def _allocate_pages(self, num_pages: int) -> Optional[List[int]]:
        if num_pages > len(self._free):
            return None                                                     # kvcache.py:60-61
        if self.ring is not None:
            if self._ring_defer is None:
                self.ring.pop(num_pages)
            else:
                self._ring_defer += num_pages
        return [self._free.popleft() for _ in range(num_pages)]

    def take_on_device(self, num_pages: int) -> bool:
        """The decode metadata kernel takes `num_pages` off the device ring itself: only the host mirror moves here.
        False (nothing taken) when the pool is short."""
        if num_pages > len(self._free):
            return False
        for _ in range(num_pages):
            self._free.popleft()
        return True

    def _free_pages(self, pages: Iterable[int]) -> None:
        if isinstance(pages, torch.Tensor):
            pages = pages.tolist()
        pages = [int(p) for p in pages]
        self._free.extend(pages)
        if self.ring is not None:
            self.ring.push(pages)

    def clear(self) -> None:
        self._free = deque(range(self.page_num))
        if self.ring is not None:
            self.ring.reset()

    def set_free_pages(self, pages: Iterable[int]) -> None:
        """Replace the FIFO content, front to back (benchmarks: shuffled physical pages)."""
        self._free = deque(int(p) for p in pages)
        if self.ring is not None:
            self.ring.load(list(self._free))

    def available_page_num(self) -> int:
        return len(self._free)

    # ---- data path
    def write_kv(self, layer_id: int, loc: torch.Tensor, cache_k: torch.Tensor, cache_v: torch.Tensor) -> None:
        """k_cache[layer][loc] = cache_k; v_cache[layer][loc] = cache_v (kvcache.py:69-79), one HIP launch."""
        ops.kv_store(self.k_cache[layer_id], self.v_cache[layer_id], loc, cache_k, cache_v)

    def get_kv_cache(self, layer_id: int) -> Tuple[torch.Tensor, torch.Tensor]:
        return self.k_cache[layer_id], self.v_cache[layer_id]
