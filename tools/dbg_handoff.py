#!/usr/bin/env python3
"""Dev: share-kernel launches on a few batches: per-request max error against torch fp32, the plan's error word."""
import os, sys, time, random
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "mini-sglang_amd"), os.path.join(REPO, "tests"), REPO, os.path.join(REPO, "tests", "golden")]
from mi355_attn import ops
from test_gpu_parity import _gpu_decode_reference
DEV = "cuda"
def case(Hq, Hkv, D, ps, lens, name):
    B = len(lens)
    g = torch.Generator(device=DEV).manual_seed(0)
    npages = sum(-(-l // ps) for l in lens)
    perm = torch.randperm(npages, generator=torch.Generator().manual_seed(1)).to(torch.int32)
    W = max(-(-l // ps) for l in lens)
    table = torch.zeros((B, W), dtype=torch.int32)
    off = 0
    for b, l in enumerate(lens):
        n = -(-l // ps)
        table[b, :n] = perm[off:off + n]
        off += n
    kp = torch.randn((npages * ps, Hkv, D), generator=g, device=DEV).to(torch.bfloat16)
    vp = torch.randn((npages * ps, Hkv, D), generator=g, device=DEV).to(torch.bfloat16)
    q = (torch.randn((B, Hq * D), generator=g, device=DEV) * 3).to(torch.bfloat16)
    dt, dr, dl = table.to(DEV), torch.arange(B, device=DEV), torch.tensor(lens, device=DEV)
    ref = _gpu_decode_reference(q, kp, vp, dt, list(range(B)), lens, ps, Hq)
    plan = ops.decode_plan(dt, dr, dl, ps, Hq, Hkv, D, sum(lens))
    share = ops.decode_planned_workspace_bytes(B, Hq, Hkv, D, sum(lens)) == 0
    hdr = plan.buf.view(torch.int32)[:8].tolist()
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.time()
        o = ops.decode_attn_planned(q, kp, vp, plan, Hq)
        torch.cuda.synchronize()
        err = (o.float() - ref).abs().view(B, Hq, D).amax(dim=2)          # [B, Hq]
        bad = (err > 0.05).nonzero().tolist()
        print(f"{name} share={share} hdr={hdr} rep {rep}: {1e3 * (time.time() - t0):.2f} ms  max diff {float(err.max()):.4f}  error word {plan.error()}  bad (request, head): {bad[:24]}{' ...' if len(bad) > 24 else ''}", flush=True)
random.seed(0)
cfg2 = [random.randint(64, 2048) + 1 for _ in range(64)]
case(32, 8, 64, 16, cfg2, "cfg2")
case(32, 8, 128, 16, cfg2, "8b")
case(8, 8, 64, 16, [70001, 64000, 59999, 61234, 66000, 63000, 58000, 65536], "long G1")
case(32, 8, 64, 16, [9000] * 32, "multi-pub G4")
