"""Paged radix tree: token ids -> physical page numbers, with the reference's interface and results
(memory/radix_cache.py:73-371).  It is the producer of `req.prefix_ppns`, hence of `prefix_lens`.

Host-only on purpose: the tree is pointer chasing over a few hundred nodes, and everything it produces is
consumed by the host planner (mi355_attn.batch.plan_prefill).  Unlike the reference, node values are Python
int lists rather than device int32 tensors, so matching never launches a kernel or syncs; the LRU stamp is
a strictly increasing counter (ties in `time.time()` made the reference's eviction order arbitrary).
One deliberate difference in results: after `match_prefix` splits a node the walk STOPS (the key is exhausted at the
split point).  The reference goes on from the new upper node without consuming the key (:263-267), which with a repeated
pattern ("abab" cached, "abc" queried at page_size 1) descends into the split-off tail and returns pages of tokens
that were never asked for - more pages than the key has.  `reference_split_walk=True` restores that walk; only the
parity tests against the reference's own traces use it.
"""
from __future__ import annotations

import heapq
import itertools
from typing import Dict, List, Optional, Tuple



class TreeNode:
    _ids = itertools.count()
    _clock = itertools.count(1)

    def __init__(self):
        self.children: Dict[object, "TreeNode"] = {}
        self.parent: Optional["TreeNode"] = None
        self.key: List[int] = []
        self.value: List[int] = []
        self.lock_ref = 0
        self.last_access_time = next(TreeNode._clock)
        self.id = next(TreeNode._ids)

    def touch(self) -> None:
        self.last_access_time = next(TreeNode._clock)

    def __lt__(self, other: "TreeNode") -> bool:
        return self.last_access_time < other.last_access_time


class PagedRadixCache:
    def __init__(self, page_manager, kvcache, page_size: Optional[int] = None, reference_split_walk: bool = False):
        self.page_manager = page_manager
        self.kvcache = kvcache
        self.page_size = page_size if page_size is not None else page_manager.page_size
        self.reference_split_walk = bool(reference_split_walk)
        self.reset()

    # ---- keys (radix_cache.py:52-70, 85-90)
    def _child_key(self, key):
        return key[0] if self.page_size == 1 else tuple(key[: self.page_size])

    def _match_len(self, a, b) -> int:
        ps = self.page_size
        n = 0
        if ps == 1:
            for x, y in zip(a, b):
                if x != y:
                    break
                n += 1
            return n
        lim = min(len(a), len(b))
        while n < lim and a[n:n + ps] == b[n:n + ps]:
            n += ps
        return n

    def reset(self) -> None:
        self.root_node = TreeNode()
        self.root_node.lock_ref = 1
        self.evictable_size_ = 0
        self.protected_size_ = 0

    # ---- lookup / insert (radix_cache.py:101-143)
    def match_prefix(self, key: List[int]) -> Tuple[List[int], TreeNode]:
        key = list(key)
        if not key:
            return [], self.root_node
        if self.page_size != 1:
            key = key[: len(key) // self.page_size * self.page_size]
        chunks, node = self._match_walk(self.root_node, key)
        return [p for c in chunks for p in c], node

    def insert(self, key: List[int], value=None) -> int:
        assert value is not None, "value must be the page numbers of `key`"
        key, value = list(key), [int(v) for v in value]
        if self.page_size != 1:
            full = len(key) // self.page_size
            key, value = key[: full * self.page_size], value[:full]
        return self._insert_walk(self.root_node, key, value)

    def _row_pages(self, row: int, lo: int, hi: int) -> List[int]:
        if hi <= lo:
            return []
        return self.page_manager.page_table[row, lo:hi].tolist()

    # ---- request life cycle (radix_cache.py:145-193)
    def cache_finished_req(self, req) -> None:
        ps = self.page_size
        token_ids = (req.origin_input_ids + req.output_ids)[:-1]
        full, ceil = len(token_ids) // ps, -(-len(token_ids) // ps)
        row = self._row_pages(req.page_table_id, 0, ceil)
        ppns, tail = row[:full], row[full:]
        if tail:
            self.kvcache._free_pages(tail)
        shared = self.insert(token_ids[: full * ps], ppns) // ps
        self.kvcache._free_pages(ppns[len(req.prefix_ppns): shared])
        self.page_manager.free([req.page_table_id])
        self.dec_lock_ref(req.last_node)

    def cache_unfinished_req(self, req) -> None:
        ps = self.page_size
        ids = req.fill_ids
        full = len(ids) // ps
        ppns = self._row_pages(req.page_table_id, 0, full)
        aligned = ids[: full * ps]
        shared = self.insert(aligned, ppns) // ps
        self.kvcache._free_pages(ppns[len(req.prefix_ppns): shared])
        new_ppns, new_node = self.match_prefix(aligned)
        n_old = len(req.prefix_ppns)
        if len(new_ppns) > n_old:
            self.page_manager.write_ppns_decode(req.page_table_id, list(range(n_old, len(new_ppns))), new_ppns[n_old:])
        self.dec_lock_ref(req.last_node)
        self.inc_lock_ref(new_node)
        req.prefix_ppns = new_ppns
        req.last_node = new_node

    # ---- eviction and pinning (radix_cache.py:202-252)
    def evict(self, num_pages) -> None:
        leaves = self._collect_leaves()
        heapq.heapify(leaves)
        done = 0
        while done < num_pages and leaves:
            x = heapq.heappop(leaves)
            if x is self.root_node:
                break
            if x.lock_ref > 0:
                continue
            if self.kvcache is not None:
                self.kvcache._free_pages(x.value)
            self._delete_leaf(x)
            done += len(x.value)
            if not x.parent.children:
                heapq.heappush(leaves, x.parent)

    def inc_lock_ref(self, node: TreeNode) -> int:
        delta = 0
        while node is not self.root_node:
            if node.lock_ref == 0:
                n = len(node.value) * self.page_size
                self.evictable_size_ -= n
                self.protected_size_ += n
                delta -= n
            node.lock_ref += 1
            node = node.parent
        return delta

    def dec_lock_ref(self, node: TreeNode) -> int:
        delta = 0
        while node is not self.root_node:
            if node.lock_ref == 1:
                n = len(node.value) * self.page_size
                self.evictable_size_ += n
                self.protected_size_ -= n
                delta += n
            node.lock_ref -= 1
            node = node.parent
        return delta

    def evictable_size(self) -> int:
        return self.evictable_size_

    def protected_size(self) -> int:
        return self.protected_size_

    def total_size(self) -> int:
        total, stack = 0, [self.root_node]
        while stack:
            n = stack.pop()
            total += len(n.value) * self.page_size
            stack.extend(n.children.values())
        return total

    # ---- walks (radix_cache.py:255-321)
    def _match_walk(self, node: TreeNode, key: List[int]):
        node.touch()
        chunks: List[List[int]] = []
        ck = self._child_key(key)
        while key and ck in node.children:
            child = node.children[ck]
            child.touch()
            n = self._match_len(child.key, key)
            if n < len(child.key):
                # The match ends inside this child: split it there and stop.  (The reference, :263-267, goes on from
                # the new upper node WITHOUT consuming the key, so `ck` is looked up among that node's children next;
                # a repeated pattern makes it descend into the split-off tail and over-report.  Kept behind
                # reference_split_walk for the parity tests.)
                node = self._split(child, n)
                chunks.append(node.value)
                if not self.reference_split_walk:
                    break
            else:
                chunks.append(child.value)
                node = child
                key = key[n:]
                if key:
                    ck = self._child_key(key)
        return chunks, node

    def _split(self, child: TreeNode, n: int) -> TreeNode:
        top = TreeNode()
        top.children = {self._child_key(child.key[n:]): child}
        top.parent = child.parent
        top.lock_ref = child.lock_ref
        top.key, top.value = child.key[:n], child.value[: n // self.page_size]
        top.parent.children[self._child_key(child.key)] = top
        child.parent = top
        child.key, child.value = child.key[n:], child.value[n // self.page_size:]
        return top

    def _insert_walk(self, node: TreeNode, key: List[int], value: List[int]) -> int:
        node.touch()
        if not key:
            return 0
        ck = self._child_key(key)
        matched = 0
        while key and ck in node.children:
            node = node.children[ck]
            node.touch()
            n = self._match_len(node.key, key)
            matched += n
            key, value = key[n:], value[n // self.page_size:]
            if n < len(node.key):
                node = self._split(node, n)
            if key:
                ck = self._child_key(key)
        if key:
            leaf = TreeNode()
            leaf.parent, leaf.key, leaf.value = node, key, value
            node.children[ck] = leaf
            self.evictable_size_ += len(value) * self.page_size
        return matched

    def _delete_leaf(self, node: TreeNode) -> None:
        for k, v in list(node.parent.children.items()):
            if v is node:
                del node.parent.children[k]
                break
        self.evictable_size_ -= len(node.key)

    def _collect_leaves(self) -> List[TreeNode]:
        out, stack = [], [self.root_node]
        while stack:
            n = stack.pop()
            if not n.children:
                out.append(n)
            else:
                stack.extend(n.children.values())
        return out


# ------------------------------------------------------------------------------------------------
# native tree: same interface, the walks run in libmi355radix.so (include/mi355radix.h)
# ------------------------------------------------------------------------------------------------
_RADIX_LIB = None


def _radix_lib():
    global _RADIX_LIB
    if _RADIX_LIB is None:
        import ctypes
        import os
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lib", "libmi355radix.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: build it (make -C mini-sglang_amd/csrc_host, or __graft_entry__.build())")
        lib = ctypes.CDLL(path)
        i64, p = ctypes.c_int64, ctypes.c_void_p
        for name, res, args in (
                ("m355r_version", ctypes.c_int, []),
                ("m355r_create", p, [ctypes.c_int32]), ("m355r_destroy", None, [p]), ("m355r_reset", None, [p]),
                ("m355r_set_reference_walk", None, [p, ctypes.c_int32]),
                ("m355r_match_prefix", i64, [p, p, i64, p, i64, p]),
                ("m355r_last_match", i64, [p, p, i64]),
                ("m355r_insert", i64, [p, p, i64, p, i64]),
                ("m355r_inc_lock_ref", i64, [p, i64]), ("m355r_dec_lock_ref", i64, [p, i64]),
                ("m355r_evict", i64, [p, i64, p, i64]),
                ("m355r_evictable_size", i64, [p]), ("m355r_protected_size", i64, [p]),
                ("m355r_total_size", i64, [p]), ("m355r_node_count", i64, [p])):
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _RADIX_LIB = lib
    return _RADIX_LIB


class NativeRadixCache(PagedRadixCache):
    """PagedRadixCache with the tree in native code.  Nodes are opaque integer handles (`req.last_node` holds one);
    the request life-cycle methods are inherited and call the primitives below."""

    def __init__(self, page_manager, kvcache, page_size: Optional[int] = None, reference_split_walk: bool = False):
        import numpy as np
        self._np = np
        self._lib = _radix_lib()
        ps = page_size if page_size is not None else page_manager.page_size
        self._tree = self._lib.m355r_create(int(ps))
        self._lib.m355r_set_reference_walk(self._tree, int(bool(reference_split_walk)))
        super().__init__(page_manager, kvcache, page_size, reference_split_walk)

    def __del__(self):
        tree, self._tree = getattr(self, "_tree", None), None
        if tree:
            self._lib.m355r_destroy(tree)

    def reset(self) -> None:
        import ctypes
        self._lib.m355r_reset(self._tree)
        node = ctypes.c_int64(0)
        self._lib.m355r_match_prefix(self._tree, None, 0, None, 0, ctypes.byref(node))
        self.root_node = int(node.value)

    def match_prefix(self, key):
        import ctypes
        np = self._np
        k = np.ascontiguousarray(np.asarray(key, dtype=np.int64))
        cap = k.size // self.page_size
        out = np.empty(max(cap, 1), dtype=np.int32)
        node = ctypes.c_int64(0)
        n = self._lib.m355r_match_prefix(self._tree, k.ctypes.data, k.size, out.ctypes.data, out.size, ctypes.byref(node))
        if n > out.size:      # only with reference_split_walk: that walk can report more pages than the key has
            out = np.empty(n, dtype=np.int32)
            self._lib.m355r_last_match(self._tree, out.ctypes.data, out.size)
        return out[:n].tolist(), int(node.value)

    def insert(self, key, value=None) -> int:
        assert value is not None, "value must be the page numbers of `key`"
        np = self._np
        k = np.ascontiguousarray(np.asarray(key, dtype=np.int64))
        v = np.ascontiguousarray(np.asarray(value, dtype=np.int32))
        return int(self._lib.m355r_insert(self._tree, k.ctypes.data, k.size, v.ctypes.data, v.size))

    def evict(self, num_pages) -> None:
        np = self._np
        out = np.empty(max(int(self._lib.m355r_total_size(self._tree)) // self.page_size, 1), dtype=np.int32)
        n = int(self._lib.m355r_evict(self._tree, int(num_pages), out.ctypes.data, out.size))
        if n > 0 and self.kvcache is not None:
            self.kvcache._free_pages(out[:n].tolist())

    def inc_lock_ref(self, node) -> int:
        return int(self._lib.m355r_inc_lock_ref(self._tree, int(node)))

    def dec_lock_ref(self, node) -> int:
        return int(self._lib.m355r_dec_lock_ref(self._tree, int(node)))

    def evictable_size(self) -> int:
        return int(self._lib.m355r_evictable_size(self._tree))

    def protected_size(self) -> int:
        return int(self._lib.m355r_protected_size(self._tree))

    def total_size(self) -> int:
        return int(self._lib.m355r_total_size(self._tree))

    def node_count(self) -> int:
        return int(self._lib.m355r_node_count(self._tree))
