#!/usr/bin/env python3
"""Dev tool: per-wave phase cycle sums of prefill_wg_kernel (8B shape, D=128, G=4; s_memtime cycles).

Phases per KV pair: 0 DMA wait + barrier, 1 issue of the next pair's DMA, 2 K reads + 16 QK^T MFMAs issued,
3 mask / max / rescale decision (first use of S: the MFMA results are waited for here), 4 exp2 + pack + 16 PV MFMAs,
5 loop overhead between pairs.  usage: stamp_prefill.py [suffix_len] [n_requests]"""
import ctypes
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "mini-sglang_amd"))
os.environ.setdefault("M355_LIB_PATH", "dev")      # timing probes live in the dev build (make -C mini-sglang_amd/csrc dev)
from mi355_attn import _lib, ops  # noqa: E402

dev = "cuda:0"
Hq, Hkv, D, ps = 32, 8, 128, 16
E = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
NW = 8 if os.environ.get("M355_PREFILL_WG") == "82" else 4
pages = B * (E // ps) + 4
perm = torch.randperm(pages, generator=torch.Generator().manual_seed(0))
table = perm[: B * (E // ps)].reshape(B, E // ps).to(torch.int32).to(dev)
g = torch.Generator(device=dev).manual_seed(0)
kp = torch.randn((pages * ps, Hkv, D), generator=g, device=dev).to(torch.bfloat16)
vp = torch.randn((pages * ps, Hkv, D), generator=g, device=dev).to(torch.bfloat16)
q = torch.randn((B * E, Hq * D), generator=g, device=dev).to(torch.bfloat16)
cu = torch.arange(B + 1, dtype=torch.int32, device=dev) * E
seq = torch.full((B,), E, device=dev, dtype=torch.int64)
pre = torch.zeros((B,), device=dev, dtype=torch.int64)
rows = torch.arange(B, device=dev)
out = torch.empty_like(q)
tiles_max = (E * (Hq // Hkv) + 32 * NW - 1) // (32 * NW)
nblocks = B * Hkv * tiles_max
stamps = torch.zeros((nblocks * NW, 16), dtype=torch.int64, device=dev)
lib = _lib.load()
lib.m355dbg_set_prefill_stamp_buffer.argtypes = [ctypes.c_void_p]


def run():
    ops.prefill_attn(q, kp, vp, table, rows, seq, pre, cu, ps, Hq, E, E, out=out)


run(); run()
torch.cuda.synchronize()
assert lib.m355dbg_set_prefill_stamp_buffer(stamps.data_ptr()) == 0
run()
torch.cuda.synchronize()
lib.m355dbg_set_prefill_stamp_buffer(None)
s = stamps.cpu().numpy().astype(np.float64)
live = s[:, 7] > 0
a = s[live]
names = ["dma wait+barrier", "slots of next pair", "K reads + QK issue", "K DMA + mask/max/decide", "V DMA + exp + PV (+drain)"]
pairs = a[:, 7]
print(f"waves {live.sum()}  pairs/wave median {np.median(pairs):.0f}  lifetime median {np.median(a[:, 6]):.0f} cyc")
tot = a[:, :5].sum(axis=1)
long = a[:, 5] > 1000
print(f"in-kernel clock: {np.median(a[long, 6] / a[long, 5]) * 100:.0f} MHz (s_memtime / s_memrealtime, waves living > 10 us)")
for sel_name, sel in (("all waves", np.ones(len(a), bool)), ("longest quarter", pairs >= np.percentile(pairs, 75))):
    print(f"-- {sel_name}: cycles per pair (sum over the wave / pairs of the workgroup)")
    for i, n in enumerate(names):
        print(f"   {n:20s} {np.mean(a[sel, i] / pairs[sel]):8.0f}")
    print(f"   {'total':20s} {np.mean(tot[sel] / pairs[sel]):8.0f}   (lifetime/pairs {np.mean(a[sel, 6] / pairs[sel]):.0f})")
