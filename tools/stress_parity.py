#!/usr/bin/env python3
"""Randomised parity sweep on the GPU (dev): planned decode (+fused store), self-contained decode and prefill against a torch
fp32 reference on the same device, over random head shapes, page sizes, batch compositions and prefix splits."""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "mini-sglang_amd"))
from mi355_attn import ops  # noqa: E402

DEV = "cuda:0"
TOL = 2e-2
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
DT = torch.float16 if (len(sys.argv) > 3 and sys.argv[3] == "f16") else torch.bfloat16     # element type of q / K / V / out


def ref_attn(q_rows, kp, vp, table, rows, seq, pre, ps, Hq):
    Hkv, D = kp.shape[1], kp.shape[2]
    outs, start = [], 0
    for b, (L, P) in enumerate(zip(seq, pre)):
        E = L - P
        pos = torch.arange(L, device=DEV)
        slots = table[rows[b], pos // ps].long() * ps + pos % ps
        k = kp[slots].float().repeat_interleave(Hq // Hkv, dim=1)
        v = vp[slots].float().repeat_interleave(Hq // Hkv, dim=1)
        qq = q_rows[start:start + E].view(E, Hq, D).float()
        s = torch.einsum("ehd,lhd->hel", qq, k) * D ** -0.5
        mask = pos[None, :] > (P + torch.arange(E, device=DEV))[:, None]
        s = s.masked_fill(mask[None], float("-inf"))
        outs.append(torch.einsum("hel,lhd->ehd", torch.softmax(s, dim=-1), v).reshape(E, Hq * D))
        start += E
    return torch.cat(outs)


t0 = time.time()
n = 0
while time.time() - t0 < budget:
    Hkv = int(rng.choice([1, 2, 4, 8]))
    G = int(rng.choice([1, 2, 4, 8]))
    D = int(rng.choice([64, 128]))
    Hq = Hkv * G
    ps = int(rng.choice([1, 3, 4, 12, 16, 64]))
    B = int(rng.choice([1, 2, 7, 33, 150]))
    mode = rng.choice(["short", "mixed", "long"])
    hi = {"short": 40, "mixed": 700, "long": 3000}[mode]
    seq = [int(x) for x in rng.integers(1, hi + 1, size=B)]
    if B > 30:
        seq = [min(s, 400) for s in seq]
    npages = sum(-(-l // ps) for l in seq) + 2
    perm = rng.permutation(npages)
    W = max(-(-l // ps) for l in seq) + 1
    table = np.full((B + 1, W), npages - 1, dtype=np.int32)
    rows = rng.permutation(B + 1)[:B]
    it = iter(perm.tolist())
    for b, l in enumerate(seq):
        for j in range(-(-l // ps)):
            table[rows[b], j] = next(it)
    g = torch.Generator(device=DEV).manual_seed(int(rng.integers(1 << 30)))
    kp = torch.randn((npages * ps, Hkv, D), generator=g, device=DEV).to(DT)
    vp = torch.randn((npages * ps, Hkv, D), generator=g, device=DEV).to(DT)
    dt, dr = torch.from_numpy(table).to(DEV), torch.from_numpy(rows.astype(np.int64)).to(DEV)
    dl = torch.tensor(seq, device=DEV)
    tag = f"Hq{Hq} Hkv{Hkv} D{D} ps{ps} B{B} {mode}"
    # ---- decode: fused store + planned, against store-then-reference
    qkv = (torch.randn((B, (Hq + 2 * Hkv) * D), generator=g, device=DEV) * 1.5).to(DT)
    q, k, v = qkv.split([Hq * D, Hkv * D, Hkv * D], dim=-1)
    loc = ops.out_cache_loc(dt, dr, dl - 1, ps)
    kp2, vp2 = kp.clone(), vp.clone()
    kp2[loc] = k.reshape(B, Hkv, D)
    vp2[loc] = v.reshape(B, Hkv, D)
    ref = ref_attn(q.contiguous(), kp2, vp2, dt, rows.tolist(), seq, [s - 1 for s in seq], ps, Hq)
    plan = ops.decode_plan(dt, dr, dl, ps, Hq, Hkv, D, sum(seq))
    kp3, vp3 = kp.clone(), vp.clone()
    o = ops.decode_attn_planned(q, kp3, vp3, plan, Hq, k_new=k, v_new=v)
    assert torch.equal(kp3, kp2) and torch.equal(vp3, vp2), f"fused store pools differ: {tag}"
    err = float((o.float() - ref).abs().max())
    assert torch.allclose(o.float(), ref, rtol=TOL, atol=TOL), f"planned decode {tag}: {err}"
    o = ops.decode_attn(q, kp2, vp2, dt, dr, dl, ps, Hq, max(seq))      # the self-contained call (its own plan)
    assert torch.allclose(o.float(), ref, rtol=TOL, atol=TOL), f"self-contained decode {tag}"
    # ---- prefill over random page-aligned prefixes
    pre = [int(rng.integers(0, l)) // ps * ps for l in seq]
    ext = [s - p for s, p in zip(seq, pre)]
    T = sum(ext)
    qf = (torch.randn((T, Hq * D), generator=g, device=DEV) * 1.5).to(DT)
    cu = torch.tensor(np.concatenate([[0], np.cumsum(ext)]), dtype=torch.int32, device=DEV)
    ref = ref_attn(qf, kp, vp, dt, rows.tolist(), seq, pre, ps, Hq)
    o = ops.prefill_attn(qf, kp, vp, dt, dr, dl, torch.tensor(pre, device=DEV), cu, ps, Hq, max(ext), max(seq))
    err = float((o.float() - ref).abs().max())
    assert torch.allclose(o.float(), ref, rtol=TOL, atol=TOL), f"prefill {tag}: {err}"
    n += 1
    print(f"ok {n:3d} {tag}  sum_len {sum(seq)}", flush=True)
print(f"{n} random configurations passed in {time.time() - t0:.0f} s")
