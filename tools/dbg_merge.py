#!/usr/bin/env python3
"""Range-kernel decode against a torch fp32 reference, per request (debug aid for the in-kernel merge).
usage (GPU box): python tools/dbg_merge.py [Hq,Hkv,D] [len,len,...] [repeats]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "mini-sglang_amd"), os.path.join(REPO, "tests"), os.path.join(REPO, "tests", "golden")):
    sys.path.insert(0, p)
import numpy as np, torch
import mi355_attn as m
from mi355_attn import ops
from test_gpu_parity import _random_paged_case, _gpu_decode_reference
DEV = "cuda:0"
Hq, Hkv, D = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "8,8,128").split(",")]
lens = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "73,36,21").split(",")]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
ps, B = 16, len(lens)
table, rows, kp, vp = _random_paged_case(B, Hq, Hkv, D, ps, lens, seed=7)
q = (torch.randn((B, Hq * D), generator=torch.Generator().manual_seed(4)) * 2).to(torch.bfloat16)
dq, dk, dv = q.to(DEV), kp.to(DEV), vp.to(DEV)
dt, dr, dl = torch.from_numpy(table).to(DEV), torch.from_numpy(rows).to(DEV), torch.tensor(lens, device=DEV)
ref = _gpu_decode_reference(dq, dk, dv, dt, rows.tolist(), lens, ps, Hq)
print("workspace bytes", ops.decode_planned_workspace_bytes(B, Hq, Hkv, D, sum(lens)))
for it in range(reps):
    plan = ops.decode_plan(dt, dr, dl, ps, Hq, Hkv, D, sum(lens))
    for layer in range(2):
        o = ops.decode_attn_planned(dq, dk, dv, plan, Hq)
        torch.cuda.synchronize()
        err = (o.float() - ref).abs().view(B, Hq, D)
        bad = [(b, round(float(err[b].max()), 4)) for b in range(B) if float(err[b].max()) > 0.03 or not torch.isfinite(err[b]).all()]
        print("rep", it, "layer", layer, "max err", float(err.max()), "bad requests", bad[:10], "plan error", plan.error())
        if bad:
            b = bad[0][0]
            print("  per head max err of request", b, [round(float(err[b, h].max()), 3) for h in range(Hq)])
# ---- dump the plan: header, records of the first ranges, arrival counts
w = plan.buf.view(torch.int32).cpu().numpy()
total, R, Bp, nwh, nr = [int(x) for x in w[:5]]
cap = plan.capacity
max_ranges = nwh + cap // 1024 + 2
arr = (8 + 3 * (B + 1) + 3) & ~3
rec = (arr + (max_ranges + B) * Hkv + 3) & ~3
print("hdr total", total, "R", R, "B", Bp, "nwh", nwh, "ranges", nr, "| arrivals at", arr, "records at", rec)
print("cum", w[8:8 + B + 1], "pbase", w[8 + B + 1: 8 + 2 * (B + 1)])
for r in range(min(nr, 4)):
    e = w[rec + 56 * r: rec + 56 * (r + 1)]
    print("range", r, "n", e[0], "np", e[1], "pieces", e[4:4 + 4 * max(e[1], 0)].reshape(-1, 4).tolist(), "ends", e[36], "cuts", e[37:37 + 2 * max(e[1], 0)].reshape(-1, 2).tolist())
nz = np.nonzero(w[arr: arr + (max_ranges + B) * Hkv])[0]
print("non-zero arrival counts", [(int(i) // Hkv, int(i) % Hkv, int(w[arr + i])) for i in nz[:32]])
bb = B - 1
print("o  ", o[bb, :8].float().tolist(), "\nref", ref[bb, :8].tolist())
print("o   head 1", o[bb, D:D + 4].float().tolist(), "\nref head 1", ref[bb, D:D + 4].tolist())
print("ratio", (o[bb].float() / ref[bb])[:16].tolist())
