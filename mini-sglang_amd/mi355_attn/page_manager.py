"""The page table of the paged KV cache: row = request slot, column = virtual page, value = physical page.

Same interface as the reference's PageManager (memory/page_manager.py:7-54) so Batch / scheduler code written for
it runs unchanged.  The table is one int32 tensor in HBM, [max_req_num, max_context_len // page_size], created zeroed;
the HIP metadata kernels scatter into it and the attention kernels index it directly (include/mi355attn.h).  A row is
not wiped when its request slot is handed back: entries past a request's live length are never dereferenced.
Request slots are handed out lowest-first and recycled in the order they were returned, like the reference's list.
"""
from __future__ import annotations

from collections import deque
from typing import Deque, Iterable, List

import torch


def _on(device, values, dtype) -> torch.Tensor:
    """Host values (list / numpy / tensor) as a device tensor of `dtype`, without a blocking copy."""
    return torch.as_tensor(values, dtype=dtype).to(device, non_blocking=True)


class PageManager:
    def __init__(self, page_size: int, max_req_num: int, max_context_len: int, device: str):
        if page_size < 1 or max_req_num < 1:
            raise ValueError("page_size and max_req_num must be positive")
        self.device = device
        self.page_size, self.max_req_num = page_size, max_req_num
        self.max_page_num = max_context_len // page_size                   # columns: pages a request can span
        self.page_table = torch.zeros(max_req_num, self.max_page_num, dtype=torch.int32, device=device)
        self._idle: Deque[int] = deque()
        self.clear()

    # ---- request slots (rows) --------------------------------------------------------------------
    @property
    def free_slots(self) -> List[int]:
        """Idle rows in hand-out order (the reference keeps this as a plain list attribute)."""
        return list(self._idle)

    def clear(self) -> None:
        self._idle = deque(range(self.max_req_num))

    def allocate(self, num_reqs: int) -> List[int]:
        if num_reqs > len(self._idle):
            raise AssertionError(f"page table has {len(self._idle)} idle rows, {num_reqs} requested")
        take = self._idle.popleft
        return [take() for _ in range(num_reqs)]

    def free(self, page_table_ids: Iterable[int]) -> None:
        for row in page_table_ids:
            self._idle.append(int(row))

    # ---- entries ---------------------------------------------------------------------------------
    def translate_token(self, req_id: int, token_id: int):
        """Reference formula as written (page_manager.py:27-29): table entry + offset inside the page."""
        page, inside = divmod(token_id, self.page_size)
        return self.page_table[req_id, page] + inside

    def write_ppns_prefill(self, page_table_id: int, ppns) -> None:
        pages = _on(self.page_table.device, ppns, torch.int32)
        self.page_table[page_table_id].narrow(0, 0, pages.numel()).copy_(pages)

    def write_ppns_decode(self, page_table_id: int, vpns, ppns) -> None:
        where = _on(self.page_table.device, vpns, torch.int64)
        self.page_table[page_table_id].index_copy_(0, where.reshape(-1), _on(self.page_table.device, ppns, torch.int32).reshape(-1))
