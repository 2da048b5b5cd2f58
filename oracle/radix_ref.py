"""Oracle: paged radix tree (token ids -> physical page numbers).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  Restates reference
``memory/radix_cache.py``; values are Python int lists instead of int32 tensors and the LRU
stamp is a logical clock (strictly increasing) instead of ``time.time()`` - the golden
generator patches the reference's clock the same way so evictions are deterministic.
"""
from __future__ import annotations

import heapq
import itertools
from typing import Dict, List, Optional, Tuple

from .memory_ref import RefPagePool, RefPageTable, RefReq

_clock = itertools.count(1)


def _tick() -> int:
    return next(_clock)


class RefNode:
    def __init__(self):
        self.children: Dict[object, "RefNode"] = {}
        self.parent: Optional["RefNode"] = None
        self.key: List[int] = []
        self.value: List[int] = []
        self.lock_ref = 0
        self.stamp = _tick()                                            # radix_cache.py:29

    def __lt__(self, other: "RefNode") -> bool:                         # :48-49
        return self.stamp < other.stamp


def match_len_tokens(a: List[int], b: List[int]) -> int:
    """page_size == 1 matcher (radix_cache.py:52-58)."""
    n = 0
    for x, y in zip(a, b):
        if x != y:
            break
        n += 1
    return n


def match_len_paged(a: List[int], b: List[int], ps: int) -> int:
    """Whole-page matcher (radix_cache.py:61-70)."""
    lim = min(len(a), len(b))
    n = 0
    while n < lim and a[n:n + ps] == b[n:n + ps]:
        n += ps
    return n


class RefRadixCache:
    def __init__(self, pt: Optional[RefPageTable], pool: Optional[RefPagePool], page_size: int):
        self.pt, self.pool, self.ps = pt, pool, page_size
        self.reset()

    # ---- key helpers (radix_cache.py:85-90)
    def _child_key(self, key: List[int]):
        return key[0] if self.ps == 1 else tuple(key[:self.ps])

    def _match(self, a: List[int], b: List[int]) -> int:
        return match_len_tokens(a, b) if self.ps == 1 else match_len_paged(a, b, self.ps)

    def reset(self) -> None:                                            # :93-99
        self.root = RefNode()
        self.root.lock_ref = 1
        self.evictable = 0
        self.protected = 0

    # ---- public API
    def match_prefix(self, key: List[int]) -> Tuple[List[int], RefNode]:   # :101-132
        key = list(key)
        if not key:
            return [], self.root
        if self.ps != 1:
            key = key[:len(key) // self.ps * self.ps]
        chunks, node = self._walk_match(self.root, key)
        return [p for c in chunks for p in c], node

    def insert(self, key: List[int], value: List[int]) -> int:          # :134-143
        key, value = list(key), list(value)
        if self.ps != 1:
            full = len(key) // self.ps
            key, value = key[:full * self.ps], value[:full]
        return self._walk_insert(self.root, key, value)

    def cache_finished_req(self, req: RefReq, token_ids: List[int], last_node: RefNode) -> None:
        """radix_cache.py:145-164; ``token_ids`` = (origin + output)[:-1]."""
        ps = self.ps
        full = len(token_ids) // ps
        ceil = -(-len(token_ids) // ps)
        row = req.page_table_id
        ppns = [int(x) for x in self.pt.table[row, :full]]
        tail = [int(x) for x in self.pt.table[row, full:ceil]]
        if tail:
            self.pool.give_back(tail)
        shared = self.insert(token_ids[:full * ps], ppns) // ps
        self.pool.give_back(ppns[len(req.prefix_ppns):shared])
        self.pt.give_rows([row])
        self.dec_lock_ref(last_node)

    def cache_unfinished_req(self, req: RefReq, last_node: RefNode) -> RefNode:
        """radix_cache.py:166-193; returns the request's new last node."""
        ps = self.ps
        ids = req.fill_ids
        full = len(ids) // ps
        row = req.page_table_id
        ppns = [int(x) for x in self.pt.table[row, :full]]
        aligned = ids[:full * ps]
        shared = self.insert(aligned, ppns) // ps
        self.pool.give_back(ppns[len(req.prefix_ppns):shared])
        new_ppns, new_node = self.match_prefix(aligned)
        n_old = len(req.prefix_ppns)
        self.pt.write_at(row, list(range(n_old, len(new_ppns))), new_ppns[n_old:])
        self.dec_lock_ref(last_node)
        self.inc_lock_ref(new_node)
        req.prefix_ppns = new_ppns
        return new_node

    def evict(self, num_pages: int) -> None:                            # :202-221
        leaves = self._leaves()
        heapq.heapify(leaves)
        done = 0
        while done < num_pages and leaves:
            x = heapq.heappop(leaves)
            if x is self.root:
                break
            if x.lock_ref > 0:
                continue
            if self.pool is not None:
                self.pool.give_back(x.value)
            self._drop_leaf(x)
            done += len(x.value)
            if not x.parent.children:
                heapq.heappush(leaves, x.parent)

    def inc_lock_ref(self, node: RefNode) -> int:                       # :223-233
        delta = 0
        while node is not self.root:
            if node.lock_ref == 0:
                n = len(node.value) * self.ps
                self.evictable -= n
                self.protected += n
                delta -= n
            node.lock_ref += 1
            node = node.parent
        return delta

    def dec_lock_ref(self, node: RefNode) -> int:                       # :235-246
        delta = 0
        while node is not self.root:
            if node.lock_ref == 1:
                n = len(node.value) * self.ps
                self.evictable += n
                self.protected -= n
                delta += n
            node.lock_ref -= 1
            node = node.parent
        return delta

    def total_size(self) -> int:                                        # :348-358
        total, stack = 0, [self.root]
        while stack:
            n = stack.pop()
            total += len(n.value) * self.ps
            stack.extend(n.children.values())
        return total

    # ---- internals
    def _walk_match(self, node: RefNode, key: List[int]):               # :255-275
        node.stamp = _tick()
        chunks: List[List[int]] = []
        ck = self._child_key(key)
        while key and ck in node.children:
            child = node.children[ck]
            child.stamp = _tick()
            n = self._match(child.key, key)
            if n < len(child.key):
                top = self._split(child, n)
                chunks.append(top.value)
                node = top
                # Faithful to the reference (:263-267): after a split neither `key` nor `ck`
                # advances and the loop does not break, so the next test looks ck (the QUERY's
                # first page) up among `top`'s children.  Normally that misses and the walk
                # ends; if the split-off remainder happens to start with the same page as the
                # query (e.g. tree "abab", query "abc" at ps=1) the walk descends into it and
                # over-reports the match.  Kept as is: parity is with the reference as written.
            else:
                chunks.append(child.value)
                node = child
                key = key[n:]
                if key:
                    ck = self._child_key(key)
        return chunks, node

    def _split(self, child: RefNode, n: int) -> RefNode:                # :277-289
        top = RefNode()
        top.children = {self._child_key(child.key[n:]): child}
        top.parent = child.parent
        top.lock_ref = child.lock_ref
        top.key = child.key[:n]
        top.value = child.value[:n // self.ps]
        child.parent = top
        old_key = child.key
        child.key = child.key[n:]
        child.value = child.value[n // self.ps:]
        top.parent.children[self._child_key(old_key)] = top
        return top

    def _walk_insert(self, node: RefNode, key: List[int], value: List[int]) -> int:   # :291-321
        node.stamp = _tick()
        if not key:
            return 0
        ck = self._child_key(key)
        matched = 0
        while key and ck in node.children:
            node = node.children[ck]
            node.stamp = _tick()
            n = self._match(node.key, key)
            matched += n
            key = key[n:]
            value = value[n // self.ps:]
            if n < len(node.key):
                node = self._split(node, n)
            if key:
                ck = self._child_key(key)
        if key:
            leaf = RefNode()
            leaf.parent = node
            leaf.key = key
            leaf.value = value
            node.children[ck] = leaf
            self.evictable += len(value) * self.ps
        return matched

    def _drop_leaf(self, node: RefNode) -> None:                        # :341-346
        for k, v in list(node.parent.children.items()):
            if v is node:
                del node.parent.children[k]
                break
        self.evictable -= len(node.key)

    def _leaves(self) -> List[RefNode]:                                 # :360-371
        out, stack = [], [self.root]
        while stack:
            n = stack.pop()
            if not n.children:
                out.append(n)
            else:
                stack.extend(n.children.values())
        return out
