#!/usr/bin/env python3
"""Dev: host time of one backend.forward (decode) call.  Tiny batch (B=1, 128 tokens) so that the GPU side (~8 us per
layer) is shorter than the host side and the loop is host-bound; prints wall us per layer call and a cProfile top list."""
import cProfile
import os
import pstats
import sys
import time
from types import SimpleNamespace

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "mini-sglang_amd"))
import mi355_attn as m  # noqa: E402

dev = "cuda:0"
Hq, Hkv, D, ps, L, B, LEN = 32, 8, 64, 16, 16, int(os.environ.get("HB", "1")), 128
kv = m.KVCache(max(4096, B * 512), ps, Hkv, D, torch.bfloat16, L, dev, device_free_list=os.environ.get("HRING") == "1")
pm = m.PageManager(ps, max(B, 8), 4096, dev)
reqs = [m.Req(i, list(range(LEN))) for i in range(B)]
for r in reqs:
    r.fill_ids = r.origin_input_ids
batch = m.Batch(reqs, pm, kv, device=dev)
batch.prepare_for_prefill()
runner = SimpleNamespace(device=dev, page_size=ps, kvcache=kv, page_manager=pm, dtype=torch.bfloat16,
                         model_config=SimpleNamespace(context_len=4096), tp_rank=0, tp_size=1)
backend = m.MI355AttnBackend(runner)
layers = [m.Attention(l, Hq, Hkv, D) for l in range(L)]
qkv = torch.randn((B, (Hq + 2 * Hkv) * D), device=dev).to(torch.bfloat16)
q, k, v = qkv.split([Hq * D, Hkv * D, Hkv * D], dim=-1)


def step():
    batch.output_ids = torch.zeros(B, dtype=torch.int64, device=dev)
    batch.prepare_for_decode()
    backend.init_forward_metadata(batch)
    for layer in layers:
        backend.forward(q=q, k=k, v=v, layer=layer, batch=batch)


for _ in range(5):
    step()
torch.cuda.synchronize()
N = 100
t0 = time.perf_counter()
for _ in range(N):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"B={B}: host {1e6 * (t1 - t0) / N:.1f} us per step = {1e6 * (t1 - t0) / N / L:.2f} us per layer call "
      f"(incl. prepare + metadata); GPU drained {1e6 * (t2 - t1):.0f} us after the loop")
pr = cProfile.Profile()
pr.enable()
for _ in range(50):
    step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumtime").print_stats("mi355_attn", 14)
