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
from harness import workloads as W  # noqa: E402

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
lo, hi = (int(x) for x in args.suffix.split(","))
case = W.prefill_cfg3_case(dev, Hq, Hkv, D, ps=args.ps, prefix=args.prefix, groups=args.groups, per_group=args.per_group,
                           suffix=(lo, hi), layers=args.layers)
r = W.time_prefill(case, reps=args.reps)
print(json.dumps({"workload": f"prefill heads={Hq},{Hkv},{D} B={case.B} prefix={case.pre[0]} extend {lo}..{hi} (T={case.T}) ps={args.ps}",
                  "us_per_layer": r["us_per_layer"], "tflops": r["tflops"], "frac_of_2.5PF": r["frac_of_mfma_peak"],
                  "gflop_per_layer": r["gflop_per_layer"], "min_bytes_MB": r["min_bytes_MB"],
                  "mode": os.environ.get("M355_PREFILL_MODE", "0")}))
