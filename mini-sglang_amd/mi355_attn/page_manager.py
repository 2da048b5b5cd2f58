"""(request row, virtual page) -> physical page table with the reference's interface
(memory/page_manager.py:7-54).  The table lives in HBM as int32 [max_req_num, max_context_len // page_size],
zero-initialised; rows are NOT cleared when a request slot is freed (stale entries beyond the live length
are never dereferenced by the kernels)."""
from __future__ import annotations

from collections import deque
from typing import List, Sequence

import torch


class PageManager:
    def __init__(self, page_size: int, max_req_num: int, max_context_len: int, device: str):
        self.page_size = page_size
        self.max_req_num = max_req_num
        self.max_page_num = max_context_len // page_size
        self.page_table = torch.zeros((self.max_req_num, self.max_page_num), dtype=torch.int32, device=device)
        self.device = device
        self.clear()

    @property
    def free_slots(self) -> List[int]:
        return list(self._free)

    def translate_token(self, req_id: int, token_id: int):
        vpn, offset = token_id // self.page_size, token_id % self.page_size
        return self.page_table[req_id, vpn] + offset                       # page_manager.py:27-29 (as written)

    def allocate(self, num_reqs: int) -> List[int]:
        assert num_reqs <= len(self._free), "No free slots in page table"
        return [self._free.popleft() for _ in range(num_reqs)]

    def free(self, page_table_ids: Sequence[int]) -> None:
        self._free.extend(int(i) for i in page_table_ids)

    def clear(self) -> None:
        self._free = deque(range(self.max_req_num))

    def write_ppns_prefill(self, page_table_id: int, ppns) -> None:
        ppns = torch.as_tensor(ppns, dtype=torch.int32).to(self.page_table.device, non_blocking=True)
        self.page_table[page_table_id, : ppns.size(0)] = ppns

    def write_ppns_decode(self, page_table_id: int, vpns, ppns) -> None:
        dev = self.page_table.device
        vpns = torch.as_tensor(vpns, dtype=torch.int64).to(dev, non_blocking=True)
        ppns = torch.as_tensor(ppns, dtype=torch.int32).to(dev, non_blocking=True)
        self.page_table[page_table_id, vpns] = ppns
