#!/usr/bin/env python3
"""Dev tool: timeline of prefill_wg_kernel on cfg3 (8B shape, 16 requests, shared 1 024-token prefix): when each row block
starts, how long its ramp (kernel entry -> first pair) and its loop take, how busy the slots are over the launch.
Needs the dev build (absolute s_memrealtime stamps, 10 ns ticks).  usage: stamp_prefill_cfg3.py [bench_prefill workload flags]"""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "mini-sglang_amd"))
os.environ.setdefault("M355_LIB_PATH", "dev")
from harness import workloads as W  # noqa: E402
from mi355_attn import _lib, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--prefix", type=int, default=1024)
ap.add_argument("--groups", type=int, default=4)
ap.add_argument("--per-group", type=int, default=4)
ap.add_argument("--suffix", default="128,512")
a = ap.parse_args()
lo, hi = (int(x) for x in a.suffix.split(","))
dev = "cuda:0"
c = W.prefill_cfg3_case(dev, 32, 8, 128, prefix=a.prefix, groups=a.groups, per_group=a.per_group, suffix=(lo, hi), layers=2)
NW = 4
tiles_max = (max(c.ext) * 4 + 32 * NW - 1) // (32 * NW)
nblocks = c.B * 8 * tiles_max
stamps = torch.zeros((nblocks * NW, 16), dtype=torch.int64, device=dev)
lib = _lib.load()
lib.m355dbg_set_prefill_stamp_buffer.argtypes = [ctypes.c_void_p]


def run(i):
    kp, vp = c.pools[i]
    ops.prefill_attn(c.q, kp, vp, c.table, c.rows, c.dseq, c.dpre, c.cu, c.ps, c.Hq, max(c.ext), max(c.seq), out=c.out)


run(0); run(1)
torch.cuda.synchronize()
assert lib.m355dbg_set_prefill_stamp_buffer(stamps.data_ptr()) == 0
run(0)
torch.cuda.synchronize()
lib.m355dbg_set_prefill_stamp_buffer(None)
s = stamps.cpu().numpy().astype(np.float64).reshape(nblocks, NW, 16)
live = s[:, 0, 7] > 0
w0 = s[live][:, 0, :]                      # wave 0 of each live block
allw = s[live]
t0 = allw[:, :, 8].min()
entry = (allw[:, :, 8].min(axis=1) - t0) * 0.01
first = (allw[:, :, 9].max(axis=1) - t0) * 0.01
end = (allw[:, :, 10].max(axis=1) - t0) * 0.01
pairs = w0[:, 7]
xcc = w0[:, 11].astype(int)
print(f"live blocks {live.sum()} of {nblocks}; span first entry -> last loop end {end.max():.1f} us; pairs/block mean {pairs.mean():.1f}")
print(f"ramp (entry -> first pair step), us: median {np.median(first - entry):.2f}  p90 {np.percentile(first - entry, 90):.2f}  max {(first - entry).max():.2f}")
loop = end - first
print(f"loop us/pair: median {np.median(loop / pairs):.2f}  p10 {np.percentile(loop / pairs, 10):.2f}  p90 {np.percentile(loop / pairs, 90):.2f}")
print(f"block life us: median {np.median(end - entry):.1f}  max {(end - entry).max():.1f}")
# occupancy over time: resident blocks in 5 us bins
edges = np.arange(0, end.max() + 5, 5.0)
occ = [(np.minimum(end, e1) - np.maximum(entry, e0)).clip(min=0).sum() / 5.0 for e0, e1 in zip(edges[:-1], edges[1:])]
print("resident blocks per 5 us bin:", " ".join(f"{o:.0f}" for o in occ))
# us/pair as a function of when the block ran (does a block alone on its CU run faster?)
for q0, q1 in ((0, 40), (40, 80), (80, 120), (120, 400)):
    sel = (entry >= q0) & (entry < q1)
    if sel.any():
        print(f"blocks entering {q0:3d}..{q1:3d} us: n {sel.sum():4d}  us/pair median {np.median(loop[sel] / pairs[sel]):.2f}  ramp median {np.median((first - entry)[sel]):.2f}")
for x in range(8):
    sel = xcc == x
    print(f"XCC {x}: blocks {sel.sum():4d}  pair-steps {pairs[sel].sum():5.0f}  last loop end {end[sel].max():6.1f} us  median us/pair {np.median(loop[sel] / pairs[sel]):.2f}")
ph = allw[:, :, :5].sum(axis=(0, 1)) / allw[:, :, 7].sum()
print("cycles per pair and wave by phase (dma wait+barrier, slots, K reads+QK issue, K DMA+max, V DMA+exp+PV):", " ".join(f"{x:.0f}" for x in ph), f"total {ph.sum():.0f}")
