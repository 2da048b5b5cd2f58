#!/usr/bin/env python3
"""Prefill (extend) attention on BASELINE.json configs[2]: Llama-3-8B head shape (Hq=32, Hkv=8, D=128), page_size 16,
4 groups x 4 requests, shared 1024-token prefix already cached, unique suffix randint(128, 512) seed 0.

Prints per-layer kernel time (HIP events over `layers` distinct pools back to back), algorithmic FLOPs
4*Hq*D*sum(E*P + E*(E+1)/2) and the fraction of the 2.5 PFLOP/s bf16 dense MFMA peak.  --tp N shards heads."""
import argparse
import json
import os
import random
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "mini-sglang_amd"))
from mi355_attn import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--heads", default="32,8,128")
ap.add_argument("--prefix", type=int, default=1024)
ap.add_argument("--groups", type=int, default=4)
ap.add_argument("--per-group", type=int, default=4)
ap.add_argument("--suffix", default="128,512")
ap.add_argument("--layers", type=int, default=8)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--ps", type=int, default=16)
args = ap.parse_args()

dev = "cuda:0"
Hq, Hkv, D = (int(x) for x in args.heads.split(","))
ps = args.ps
lo, hi = (int(x) for x in args.suffix.split(","))
random.seed(0)
B = args.groups * args.per_group
ext = [random.randint(lo, hi) for _ in range(B)]
pre = [args.prefix // ps * ps] * B
seq = [p + e for p, e in zip(pre, ext)]
T = sum(ext)
# shared prefix pages per group, own pages for the suffix
npp = pre[0] // ps
pages_total = args.groups * npp + sum(-(-s // ps) - npp for s in seq) + 4
perm = torch.randperm(pages_total, generator=torch.Generator().manual_seed(0)).tolist()
it = iter(perm)
W = max(-(-s // ps) for s in seq) + 1
table = torch.zeros((B, W), dtype=torch.int32)
for gi in range(args.groups):
    shared = [next(it) for _ in range(npp)]
    for k in range(args.per_group):
        b = gi * args.per_group + k
        own = [next(it) for _ in range(-(-seq[b] // ps) - npp)]
        table[b, : npp + len(own)] = torch.tensor(shared + own, dtype=torch.int32)
g = torch.Generator(device=dev).manual_seed(0)
pools = [(torch.randn((pages_total * ps, Hkv, D), generator=g, device=dev).to(torch.bfloat16),
          torch.randn((pages_total * ps, Hkv, D), generator=g, device=dev).to(torch.bfloat16)) for _ in range(args.layers)]
q = torch.randn((T, Hq * D), generator=g, device=dev).to(torch.bfloat16)
cu = torch.tensor([0] + list(torch.tensor(ext).cumsum(0)), dtype=torch.int32, device=dev)
dt, dr = table.to(dev), torch.arange(B, device=dev)
dseq, dpre = torch.tensor(seq, device=dev), torch.tensor(pre, device=dev)
out = torch.empty((T, Hq * D), dtype=torch.bfloat16, device=dev)


def run():
    for kp, vp in pools:
        ops.prefill_attn(q, kp, vp, dt, dr, dseq, dpre, cu, ps, Hq, max(ext), max(seq), out=out)


run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(args.reps):
    run()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / (args.reps * args.layers)
flops = 4 * Hq * D * sum(e * p + e * (e + 1) // 2 for e, p in zip(ext, pre))
byts = 4 * Hkv * D * sum(seq) + 4 * T * Hq * D
print(json.dumps({"workload": f"prefill heads={Hq},{Hkv},{D} B={B} prefix={pre[0]} extend {lo}..{hi} (T={T}) ps={ps}",
                  "us_per_layer": round(us, 2), "tflops": round(flops / us / 1e6, 1), "frac_of_2.5PF": round(flops / us / 1e6 / 2500, 4),
                  "gflop_per_layer": round(flops / 1e9, 2), "min_bytes_MB": round(byts / 1e6, 2),
                  "mode": os.environ.get("M355_PREFILL_MODE", "0")}))
