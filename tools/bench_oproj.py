#!/usr/bin/env python3
"""Dev tool: GPU time of m355_oproj_allreduce with a world of one (GEMM + push to its own slot + reduce) against torch.matmul,
20 calls replayed from a HIP graph, over T and K.  `python tools/bench_oproj.py [T,T,...] [K,K,...]`"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "mini-sglang_amd"))
from mi355_attn.tp import FusedRowParallel  # noqa: E402

dev = "cuda:0"
Ts = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "16,64,256").split(",")]
Ks = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "512,2048,4096").split(",")]
f = FusedRowParallel(4096, t_cap=256, local_world=1, device=dev, device_epoch=os.environ.get("DEVICE_EPOCH", "1") != "0")
for T in Ts:
    for K in Ks:
        x = torch.randn((T, K), device=dev, dtype=torch.bfloat16)
        w = torch.randn((4096, K), device=dev, dtype=torch.bfloat16) * 0.02
        o = [torch.empty((T, 4096), device=dev, dtype=torch.bfloat16)]
        t = {}
        for name, fn in (("fused", lambda: f([x], [w], out=o)), ("matmul", lambda: torch.matmul(x, w.t(), out=o[0]))):
            fn()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(20):
                    fn()
            g.replay()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            t[name] = e0.elapsed_time(e1) * 1e3 / 100
        want = torch.matmul(x, w.t())
        err = float((o[0].float() - want.float()).abs().max())
        print(f"T={T:4d} K={K:5d}  fused {t['fused']:7.2f} us   matmul {t['matmul']:7.2f} us   weight stream at 8 TB/s {4096 * K * 2 / 8e6:5.2f} us   max diff {err:.4f}", flush=True)
f.close()
