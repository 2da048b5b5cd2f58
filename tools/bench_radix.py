#!/usr/bin/env python3
"""Host-only: the native radix tree against the Python tree on a serving-like key set (shared system prompts of
1-4k tokens + unique tails), page_size 16.  Prints microseconds per insert / match_prefix."""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "mini-sglang_amd"))
from mi355_attn.radix_cache import NativeRadixCache, PagedRadixCache  # noqa: E402


class Pool:
    def _free_pages(self, pages):
        pass


ps = 16
rng = np.random.default_rng(0)
systems = [rng.integers(0, 32000, size=int(n)).tolist() for n in rng.integers(1024, 4096, size=16)]
keys = []
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 1500):
    s = systems[int(rng.integers(0, len(systems)))]
    keys.append(s + rng.integers(0, 32000, size=int(rng.integers(64, 1024))).tolist())
pages = [list(range(len(k) // ps)) for k in keys]
akeys = [np.asarray(k, dtype=np.int64) for k in keys]
apages = [np.asarray(p, dtype=np.int32) for p in pages]
for cls, keys, pages, note in ((PagedRadixCache, keys, pages, "list keys"), (NativeRadixCache, keys, pages, "list keys"),
                               (NativeRadixCache, akeys, apages, "int64 array keys")):
    tree = cls(None, Pool(), page_size=ps)
    t0 = time.perf_counter()
    for k, p in zip(keys, pages):
        tree.insert(k, p)
    t1 = time.perf_counter()
    hit = 0
    for k in keys:
        v, node = tree.match_prefix(k)
        hit += len(v)
    t2 = time.perf_counter()
    tree.evict(10 ** 9)
    t3 = time.perf_counter()
    print(f"{cls.__name__:18s} {note:17s} insert {1e6 * (t1 - t0) / len(keys):8.1f} us  match_prefix {1e6 * (t2 - t1) / len(keys):8.1f} us  "
          f"evict-all {1e3 * (t3 - t2):7.1f} ms   (keys {len(keys)}, mean {np.mean([len(k) for k in keys]):.0f} tokens, pages hit {hit})")
