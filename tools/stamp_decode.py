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
STREAM = True        # (the round-1 chunked kernel this tool also stamped is gone: the per-wave-range kernel only)
if STREAM:
    os.environ["M355_STREAM_VARIANT"] = "0"
    os.environ["M355_STREAM_STAMP"] = "1"
os.environ.setdefault("M355_LIB_PATH", "dev")      # timing probes live in the dev build (make -C mini-sglang_amd/csrc dev)
import mi355_attn as m  # noqa: E402
from mi355_attn import _lib, ops  # noqa: E402

dev = "cuda:0"
Hq, Hkv, D, ps, L = 32, 8, 64, 16, 4
chunk = 256
random.seed(0)
lens = [random.randint(64, 2048) + 1 for _ in range(64)]
if len(sys.argv) > 1:                                  # usage: stamp_decode.py [Hq,Hkv,D fixed_len batch]
    Hq, Hkv, D = (int(x) for x in sys.argv[1].split(","))
    lens = [int(sys.argv[2])] * int(sys.argv[3])
os.environ.setdefault("M355_DECODE_SHARE", "0")        # the range kernel is what this tool stamps
B = len(lens)
pages = sum(-(-l // ps) for l in lens) + 8
kv = m.KVCache(pages * ps, ps, Hkv, D, torch.bfloat16, L, dev)
pm = m.PageManager(ps, B, max(4096, max(lens) + 64), dev)
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
nblocks = 256 * 8 if STREAM else B * Hkv * max_splits
stamps = torch.zeros((nblocks, 8), dtype=torch.int64, device=dev)
lib = _lib.load()
setter = lib.m355dbg_set_stream_stamp_buffer
setter.argtypes = [ctypes.c_void_p]
assert setter(stamps.data_ptr()) == 0
plan = ops.decode_plan(pm.page_table, batch.page_table_ids, batch.seq_lens, ps, Hq, Hkv, D, sum(lens), pool_row_stride=ops.pool_pitch(*kv.get_kv_cache(0))) if STREAM else None


def run(layer):
    kp, vp = kv.get_kv_cache(layer)
    ops.decode_attn_planned(q, kp, vp, plan, Hq)


for l in range(L):
    run(l)
torch.cuda.synchronize()
stamps.zero_()
torch.cuda.synchronize()
run(0)
torch.cuda.synchronize()
s = stamps.cpu().numpy().astype(np.int64)
live = (s[:, 5] > 0) & (s[:, 1] > 0) & (s[:, 4] > 0)      # waves that had a range (the others leave at the record)
dead = (s[:, 0] > 0) & ~live
t0 = s[s[:, 0] > 0, 0].min()
print(f"chunk {chunk}: blocks {nblocks}, live {live.sum()}, dead {dead.sum()}")
us = lambda x: x * 0.01
a = s[live]
print(f"entry time   (us after first wave): median {us(np.median(a[:,0]-t0)):.2f}  p95 {us(np.percentile(a[:,0]-t0,95)):.2f}  max {us((a[:,0]-t0).max()):.2f}")
names = ["record+slots", "(pass start)", "tiles issued+q staged", "loop", "stores retired"] if STREAM else ["lens/row", "slots+q", "first tile", "loop", "finish+stores"]
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
if STREAM:
    hdr = plan.buf.view(torch.int32)[:8].cpu().numpy()
    total, R, nr = int(hdr[0]), int(hdr[1]), int(hdr[4])
    cum = np.concatenate([[0], np.cumsum(lens)])
    nwh = 256 * 8 // Hkv
    loop = (a[:, 4] - a[:, 3]) * 0.01
    life = (a[:, 5] - a[:, 0]) * 0.01
    w = (np.arange(nblocks) // Hkv)[live]
    npieces = np.array([np.searchsorted(cum, min((x + 1) * R, total), side="left") - np.searchsorted(cum, x * R, side="right") + 1 for x in w])
    print(f"R={R} ranges={nr} total={total}")
    for k in sorted(set(npieces)):
        sel = npieces == k
        print(f"pieces={k}: waves {sel.sum():5d}  loop median {np.median(loop[sel]):.2f}  p95 {np.percentile(loop[sel],95):.2f}")
    for x in range(8):
        sel = ((np.arange(nblocks) % 8) == x)[live]
        print(f"blockIdx%8={x}: loop median {np.median(loop[sel]):.2f} p95 {np.percentile(loop[sel],95):.2f} life median {np.median(life[sel]):.2f}")
    # last waves of the stream have a short range
    print("loop time of the 16 slowest waves:", np.sort(loop)[-16:].round(1), "their w:", w[np.argsort(loop)[-16:]])
if STREAM:
    blk = np.arange(nblocks)
    head = (blk + (blk // Hkv) // max(1, nwh // Hkv)) % Hkv
    first = (s[:, 3] - s[:, 0]) * 0.01
    endt = (s[:, 5] - t0) * 0.01
    print("by kv head: first tile landed after entry (median) / loop median / end median / end max")
    for h_ in range(Hkv):
        sel = (head == h_) & live
        print(f"  h={h_}: {np.median(first[sel]):5.2f} / {np.median(loop[sel[live]] if False else ((s[sel,4]-s[sel,3])*0.01)):6.2f} / {np.median(endt[sel]):6.2f} / {endt[sel].max():6.2f}")
