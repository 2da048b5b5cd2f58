#!/usr/bin/env python3
"""Dev tool: per-wave phase timing of the decode split kernel on the cfg2 batch (needs M355_DECODE_VARIANT=17).

Stamps (s_memrealtime, 10 ns): 0 entry, 1 lens/row known, 2 slots staged + q landed, 3 first tile landed,
4 loop done, 5 partial/output stores retired."""
import ctypes
import os
import random
import sys
from collections import deque
from types import SimpleNamespace

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "mini-sglang_amd"))
os.environ.setdefault("M355_DECODE_VARIANT", "17")
import mi355_attn as m  # noqa: E402
from mi355_attn import _lib, ops  # noqa: E402

dev = "cuda:0"
Hq, Hkv, D, ps, L = 32, 8, 64, 16, 4
chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 256
random.seed(0)
lens = [random.randint(64, 2048) + 1 for _ in range(64)]
B = len(lens)
pages = sum(-(-l // ps) for l in lens) + 8
kv = m.KVCache(pages * ps, ps, Hkv, D, torch.bfloat16, L, dev)
pm = m.PageManager(ps, B, 4096, dev)
kv._free = deque(torch.randperm(pages, generator=torch.Generator().manual_seed(0)).tolist())
kv._store.copy_(torch.randn(kv._store.shape, device=dev).to(torch.bfloat16))
reqs = [m.Req(i, list(range(l))) for i, l in enumerate(lens)]
for r in reqs:
    r.fill_ids = r.origin_input_ids
batch = m.Batch(reqs, pm, kv, device=dev)
batch.prepare_for_prefill()
q = torch.randn((B, Hq * D), device=dev).to(torch.bfloat16)
max_len = max(lens)
max_splits = -(-max_len // chunk)
nblocks = B * Hkv * max_splits
stamps = torch.zeros((nblocks, 8), dtype=torch.int64, device=dev)
lib = _lib.load()
lib.m355dbg_set_stamp_buffer.argtypes = [ctypes.c_void_p]
assert lib.m355dbg_set_stamp_buffer(stamps.data_ptr()) == 0
ws = torch.empty(ops.decode_workspace_bytes(B, Hq, D, max_len, chunk), dtype=torch.uint8, device=dev)


def run(layer):
    kp, vp = kv.get_kv_cache(layer)
    ops.decode_attn(q, kp, vp, pm.page_table, batch.page_table_ids, batch.seq_lens, ps, Hq, max_len,
                    chunk_tokens=chunk, workspace=ws)


for l in range(L):
    run(l)
torch.cuda.synchronize()
stamps.zero_()
torch.cuda.synchronize()
run(0)
torch.cuda.synchronize()
s = stamps.cpu().numpy().astype(np.int64)
live = s[:, 5] > 0
dead = (s[:, 0] > 0) & ~live
t0 = s[s[:, 0] > 0, 0].min()
print(f"chunk {chunk}: blocks {nblocks}, live {live.sum()}, dead {dead.sum()}")
us = lambda x: x * 0.01
a = s[live]
print(f"entry time   (us after first wave): median {us(np.median(a[:,0]-t0)):.2f}  p95 {us(np.percentile(a[:,0]-t0,95)):.2f}  max {us((a[:,0]-t0).max()):.2f}")
names = ["lens/row", "slots+q", "first tile", "loop", "finish+stores"]
for k in range(5):
    d = a[:, k + 1] - a[:, k]
    print(f"phase {names[k]:14s}: median {us(np.median(d)):.2f} us  p95 {us(np.percentile(d,95)):.2f}  max {us(d.max()):.2f}")
print(f"wave lifetime: median {us(np.median(a[:,5]-a[:,0])):.2f} us, max {us((a[:,5]-a[:,0]).max()):.2f}")
print(f"kernel span (first entry -> last store retired): {us(a[:,5].max()-t0):.2f} us")
if dead.sum():
    print(f"dead-block entry: median {us(np.median(s[dead,0]-t0)):.2f} max {us((s[dead,0]-t0).max()):.2f}")
# how the finish time is distributed
end = np.sort(a[:, 5] - t0)
print("wave end-time percentiles (us):", [round(us(np.percentile(end, p)), 2) for p in (5, 25, 50, 75, 95, 100)])
