"""Oracle: free-page FIFO, page table and the per-batch index metadata.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  Plain Python ints + numpy; every
function restates one reference routine and says which.  All results are exact integers.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np


def ceil_div(a: int, b: int) -> int:
    return -(-a // b)


class RefPagePool:
    """Free physical pages: handed out from the FRONT, returned to the BACK
    (reference memory/kvcache.py:38-67, 87-91)."""

    def __init__(self, page_num: int, page_size: int):
        self.page_num = page_num
        self.page_size = page_size
        self.free_pages: List[int] = list(range(page_num))          # kvcache.py:87-88

    def take(self, n: int) -> Optional[List[int]]:
        if n > len(self.free_pages):                                   # kvcache.py:60-61
            return None
        got, self.free_pages = self.free_pages[:n], self.free_pages[n:]
        return got

    def give_back(self, pages: Sequence[int]) -> None:                 # kvcache.py:66-67
        self.free_pages.extend(int(p) for p in pages)

    def pages_for_prefill(self, seq_lens: Sequence[int], prefix_lens: Sequence[int]):
        """ceil(seq/ps) - ceil(prefix/ps) fresh pages per request (kvcache.py:38-47)."""
        ps = self.page_size
        return [self.take(ceil_div(int(s), ps) - ceil_div(int(p), ps))
                for s, p in zip(seq_lens, prefix_lens)]

    def pages_for_decode(self, seq_lens: Sequence[int]):
        """Same with prefix = seq-1: one page iff (seq-1) % ps == 0 (kvcache.py:49-56)."""
        return self.pages_for_prefill(seq_lens, [int(s) - 1 for s in seq_lens])

    def available(self) -> int:
        return len(self.free_pages)


class RefPageTable:
    """(request row, virtual page) -> physical page, int32, zero-initialised, rows never
    cleared on free (reference memory/page_manager.py:12-54)."""

    def __init__(self, page_size: int, max_req_num: int, max_context_len: int):
        self.page_size = page_size
        self.table = np.zeros((max_req_num, max_context_len // page_size), dtype=np.int32)
        self.free_rows: List[int] = list(range(max_req_num))

    def take_rows(self, n: int) -> List[int]:                          # page_manager.py:31-36
        assert n <= len(self.free_rows)
        got, self.free_rows = self.free_rows[:n], self.free_rows[n:]
        return got

    def give_rows(self, rows: Sequence[int]) -> None:                  # page_manager.py:38-40
        self.free_rows.extend(int(r) for r in rows)

    def write_prefix(self, row: int, ppns: Sequence[int]) -> None:     # page_manager.py:45-46
        self.table[row, :len(ppns)] = np.asarray(ppns, dtype=np.int32)

    def write_at(self, row: int, vpns: Sequence[int], ppns: Sequence[int]) -> None:  # :48-54
        self.table[row, np.asarray(vpns, dtype=np.int64)] = np.asarray(ppns, dtype=np.int32)


def out_cache_loc_ref(table: np.ndarray, rows: np.ndarray, positions: np.ndarray, page_size: int) -> np.ndarray:
    """slot = table[row, pos // ps] * ps + pos % ps, promoted to int64
    (reference engine/batch.py:315-317, 341-343)."""
    rows = np.asarray(rows, dtype=np.int64)
    positions = np.asarray(positions, dtype=np.int64)
    return table[rows, positions // page_size].astype(np.int64) * page_size + positions % page_size


@dataclass
class RefReq:
    """The request fields the metadata builders read (reference engine/batch.py:91-104)."""
    fill_ids: List[int]
    prefix_ppns: List[int] = field(default_factory=list)
    page_table_id: Optional[int] = None


@dataclass
class RefBatchMeta:
    input_ids: np.ndarray
    page_table_ids: np.ndarray
    seq_lens: np.ndarray
    prefix_lens: np.ndarray
    positions: np.ndarray
    out_cache_loc: np.ndarray


def prepare_for_prefill_ref(reqs: List[RefReq], pool: RefPagePool, pt: RefPageTable) -> RefBatchMeta:
    """Reference engine/batch.py:244-317, one request at a time."""
    ps = pt.page_size
    rows = pt.take_rows(len(reqs))                                     # :248
    seq_lens = [len(r.fill_ids) for r in reqs]                         # :255
    prefix_lens = [len(r.prefix_ppns) * ps for r in reqs]              # :256
    input_ids: List[int] = []
    for r, pl in zip(reqs, prefix_lens):                               # :251-253, :262
        input_ids.extend(r.fill_ids[pl:])
    fresh = pool.pages_for_prefill(seq_lens, prefix_lens)              # :276-278
    for r, row, pages in zip(reqs, rows, fresh):                       # :280-292
        r.page_table_id = row
        pt.write_prefix(row, list(r.prefix_ppns) + list(pages))
    positions = np.concatenate([np.arange(p, s, dtype=np.int64)        # :299-304
                                for p, s in zip(prefix_lens, seq_lens)]) if reqs else np.zeros(0, np.int64)
    row_of_token = np.concatenate([np.full(s - p, row, dtype=np.int64)  # :306-311
                                   for row, p, s in zip(rows, prefix_lens, seq_lens)]) if reqs else np.zeros(0, np.int64)
    loc = out_cache_loc_ref(pt.table, row_of_token, positions, ps)     # :315-317
    return RefBatchMeta(np.asarray(input_ids, dtype=np.int64), np.asarray(rows, dtype=np.int64),
                        np.asarray(seq_lens, dtype=np.int64), np.asarray(prefix_lens, dtype=np.int64),
                        positions, loc)


def prepare_for_decode_ref(meta: RefBatchMeta, next_ids: Sequence[int], pool: RefPagePool,
                           pt: RefPageTable) -> RefBatchMeta:
    """Reference engine/batch.py:319-343.  ``meta`` is the previous step's metadata."""
    ps = pt.page_size
    seq_lens = meta.seq_lens + 1                                       # :325
    positions = seq_lens - 1                                           # :326
    fresh = pool.pages_for_decode(seq_lens.tolist())                   # :328
    last_vpn = positions // ps                                         # :329
    for i, pages in enumerate(fresh):                                  # :331-337
        if len(pages) > 0:
            pt.write_at(int(meta.page_table_ids[i]), [int(last_vpn[i])], pages)
    loc = out_cache_loc_ref(pt.table, meta.page_table_ids, positions, ps)   # :339-343
    return RefBatchMeta(np.asarray(next_ids, dtype=np.int64), meta.page_table_ids, seq_lens,
                        meta.prefix_lens, positions.astype(np.int64), loc)
