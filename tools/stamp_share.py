#!/usr/bin/env python3
"""Dev tool: per-wave phase timing of decode_share_kernel on the cfg2 batch (M355_STREAM_STAMP build path).

Streaming waves: 0 entry, 1 share header known, 2 first records landed, 3 first stage landed, 4 last stage issued+all
but last consumed, 5 done (stores retired).  Finishing wave: 2 first segment's waves arrived, 3 hand-off complete,
4 last segment's waves arrived, 5 done."""
import ctypes
import os
import random
import sys
from collections import deque

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "mini-sglang_amd"))
os.environ["M355_STREAM_STAMP"] = "1"
os.environ.setdefault("M355_DECODE_SHARE", "1")
os.environ.setdefault("M355_LIB_PATH", "dev")      # timing probes live in the dev build (make -C mini-sglang_amd/csrc dev)
import mi355_attn as m  # noqa: E402
from mi355_attn import _lib, ops  # noqa: E402

dev = "cuda:0"
Hq, Hkv, D, ps, L = 32, 8, 64, 16, 4
random.seed(0)
lens = [random.randint(64, 2048) + 1 for _ in range(64)]
CUSTOM = "STAMP_SHAPE" in os.environ                 # e.g. STAMP_SHAPE=8,1,128 STAMP_LENS=32768x32: the summary only
if CUSTOM:
    Hq, Hkv, D = (int(x) for x in os.environ["STAMP_SHAPE"].split(","))
    n_, b_ = os.environ.get("STAMP_LENS", "32768x32").split("x")
    lens = [int(n_) + 1] * int(b_)
    L = 2
B = len(lens)
pages = sum(-(-l // ps) for l in lens) + 8
kv = m.KVCache(pages * ps, ps, Hkv, D, torch.bfloat16, L, dev, row_pad=int(os.environ['ROW_PAD']) if 'ROW_PAD' in os.environ else None)
pm = m.PageManager(ps, B, max(4096, max(lens) + ps), dev)
kv._free = deque(torch.randperm(pages, generator=torch.Generator().manual_seed(0)).tolist())
kv._store.copy_(torch.randn(kv._store.shape, device=dev).to(torch.bfloat16))
reqs = [m.Req(i, list(range(l))) for i, l in enumerate(lens)]
for r in reqs:
    r.fill_ids = r.origin_input_ids
batch = m.Batch(reqs, pm, kv, device=dev)
batch.prepare_for_prefill()
q = torch.randn((B, Hq * D), device=dev).to(torch.bfloat16)
NW = 9                              # stamp rows per workgroup: kShareMaxWaves + 1
nblocks = 256
stamps = torch.zeros((nblocks * NW, 8), dtype=torch.int64, device=dev)
lib = _lib.load()
setter = lib.m355dbg_set_stream_stamp_buffer
setter.argtypes = [ctypes.c_void_p]
assert setter(stamps.data_ptr()) == 0
plan = ops.decode_plan(pm.page_table, batch.page_table_ids, batch.seq_lens, ps, Hq, Hkv, D, sum(lens), pool_row_stride=ops.pool_pitch(*kv.get_kv_cache(0)))
WS = int(plan.buf.view(torch.int32)[4].item())      # streaming waves per workgroup; the finishing wave is wave WS


def run(layer):
    kp, vp = kv.get_kv_cache(layer)
    ops.decode_attn_planned(q, kp, vp, plan, Hq)


for l in range(L):
    run(l)
torch.cuda.synchronize()
us = lambda x: x * 0.01
for rep in range(int(os.environ.get("STAMP_REPS", "1"))):      # several launches: is the per-XCD pattern stable?
    stamps.zero_()
    torch.cuda.synchronize()
    for l in range(L):                  # back to back, as the model runs them; the LAST layer's stamps stay
        run(l)
    torch.cuda.synchronize()
    s = stamps.cpu().numpy().astype(np.int64).reshape(nblocks, NW, 8)
    t0 = s[:, :, 0][s[:, :, 0] > 0].min()
    wg_e = (s[:, :WS, 5].max(axis=1) - t0) * 0.01
    fin_e = (s[:, WS, 5] - t0) * 0.01
    first = (np.where(s[:, :WS, 3] > 0, s[:, :WS, 3], 1 << 62).min(axis=1) - t0) * 0.01
    xcc = s[:, 0, 6]
    live = s[:, 0, 5] > 0
    print(f"rep {rep}: span {max(wg_e.max(), fin_e.max()):.2f}  streaming end p50 {np.median(wg_e[live]):.2f} max {wg_e.max():.2f}  finishing end p50 {np.median(fin_e[fin_e > 0]):.2f} max {fin_e.max():.2f}")
    print("   by XCC_ID  (blocks, blockIdx%8 seen, first stage landed p50, streaming end p50 / max):")
    for x in range(8):
        sel = live & (xcc == x)
        if sel.any():
            print(f"     xcc {x}: n={int(sel.sum())} slots {sorted(set((np.nonzero(sel)[0] % 8).tolist()))}  first {np.median(first[sel]):.2f}  end {np.median(wg_e[sel]):.2f} / {wg_e[sel].max():.2f}")
st = s[:, :WS].reshape(-1, 8)
st = st[st[:, 5] > 0]
hp = s[:, WS]
hp = hp[hp[:, 5] > 0]
print(f"streaming waves {len(st)}, finishing waves {len(hp)}")
for name, a in (("streaming", st), ("finishing", hp)):
    print(f"[{name}] entry after first wave: median {us(np.median(a[:,0]-t0)):.2f} max {us((a[:,0]-t0).max()):.2f}")
    for k in range(1, 6):
        ok = a[:, k] > 0
        d = a[ok, k] - t0
        print(f"[{name}] stamp {k} (us after first entry): median {us(np.median(d)):.2f}  p5 {us(np.percentile(d,5)):.2f}  p95 {us(np.percentile(d,95)):.2f}  max {us(d.max()):.2f}   (n={ok.sum()})")
print(f"kernel span: {us(max(st[:,5].max(), hp[:,5].max()) - t0):.2f} us; last streaming wave done {us(st[:,5].max()-t0):.2f}")
tail = hp[:, 5] - hp[:, 4]
print(f"finishing wave: last arrival -> done: median {us(np.median(tail)):.2f} p95 {us(np.percentile(tail,95)):.2f} max {us(tail.max()):.2f}")
loop = st[:, 4] - st[:, 3]
print(f"streaming loop (first stage landed -> last issued): median {us(np.median(loop)):.2f} p95 {us(np.percentile(loop,95)):.2f} max {us(loop.max()):.2f}")
hdr = plan.buf.view(torch.int32)[:8].cpu().numpy()
print("plan header: total", hdr[0], "share stages", hdr[1], "shares", hdr[3], "waves", hdr[4], "stages", hdr[5], "error", hdr[6])
wg_end_ = (s[:, :WS, 5] - t0).max(axis=1) * 0.01
print("workgroup end percentiles (streaming waves)", [round(float(np.percentile(wg_end_, q)), 2) for q in (5, 25, 50, 75, 95, 100)])
fin_end = (s[:, WS, 5] - t0) * 0.01
print("finishing wave end percentiles", [round(float(np.percentile(fin_end[fin_end > 0], q)), 2) for q in (5, 25, 50, 75, 95, 100)])
if CUSTOM:
    f = s[:, WS]
    for k, name in ((2, "first segment's waves arrived"), (3, "hand-off complete"), (4, "last segment's waves arrived"), (5, "done")):
        ok = f[:, k] > 0
        if ok.any():
            d = (f[ok, k] - t0) * 0.01
            print(f"[finishing] {name}: n={ok.sum()} median {np.median(d):.2f} p5 {np.percentile(d,5):.2f} p95 {np.percentile(d,95):.2f} max {d.max():.2f}")
    if Hkv == 1:                     # long requests: how long after its LAST publisher's stores does a merger complete?
        rows = []
        for b in range(nblocks):
            if f[b, 3] == 0:
                continue
            pubs = [q for q in range(max(0, b - 16), b) if f[q, 7] > 0]
            k = 0
            while b - 1 - k >= 0 and f[b - 1 - k, 7] > 0 and k < 16:
                k += 1
            if k == 0:
                continue
            last_pub = max(f[b - 1 - i, 7] for i in range(k))
            rows.append((k, (last_pub - t0) * 0.01, (f[b, 4] - t0) * 0.01, (f[b, 3] - t0) * 0.01, (f[b, 5] - t0) * 0.01))
        for r in rows[:40]:
            print("   publishers %d: last stored %.2f | merger: own last arrival %.2f, hand-off complete %.2f, done %.2f" % r)
    sys.exit(0)
# ---- where does the spread come from: wave index, XCD slot, workgroup
S5 = (s[:, :8, 5] - t0) * 0.01
S3 = (s[:, :8, 3] - t0) * 0.01
S2 = (s[:, :8, 2] - t0) * 0.01
print("by wave index j: first stage landed (median), done (median / max)")
for j in range(8):
    print(f"  j={j}: {np.median(S3[:, j]):6.2f}   {np.median(S5[:, j]):6.2f} / {S5[:, j].max():6.2f}")
print("by blockIdx % 8: first stage landed (median), done (median / max)")
for x in range(8):
    sel = (np.arange(nblocks) % 8) == x
    print(f"  x={x}: {np.median(S3[sel]):6.2f}   {np.median(S5[sel]):6.2f} / {S5[sel].max():6.2f}")
wg_end = S5.max(axis=1)
wg_start3 = S3.min(axis=1)
print("workgroup end (max over its waves): percentiles", [round(float(np.percentile(wg_end, p)), 2) for p in (5, 25, 50, 75, 95, 100)])
print("workgroup spread (max - min wave end): median", round(float(np.median(S5.max(axis=1) - S5.min(axis=1))), 2), "max", round(float((S5.max(axis=1) - S5.min(axis=1)).max()), 2))
share_tab = plan.buf.view(torch.int32).cpu().numpy()
order = np.argsort(wg_end)
print("slowest workgroups (blockIdx, share, end, first-stage-min):", [(int(b), int(b) // 8, round(float(wg_end[b]), 1), round(float(wg_start3[b]), 1)) for b in order[-8:]])
print("fastest workgroups:", [(int(b), int(b) // 8, round(float(wg_end[b]), 1)) for b in order[:8]])
by_share = np.array([wg_end[np.arange(nblocks) // 8 == i].mean() for i in range(32)])
print("mean end by share:", by_share.round(1))
blk = np.arange(nblocks)
ROT = int(os.environ.get('M355_SHARE_ROT', '4'))
head = (blk + ((blk // 8) // ROT if ROT > 0 else 0)) % 8
print("by kv head: workgroup end median / max; streaming-wave loop median; first stage landed median")
LOOP = (s[:, :8, 4] - s[:, :8, 3]) * 0.01
for h_ in range(8):
    sel = head == h_
    print(f"  h={h_}: {np.median(wg_end[sel]):6.2f} / {wg_end[sel].max():6.2f}   loop {np.median(LOOP[sel]):6.2f}   first {np.median(S3[sel]):5.2f}")
if os.environ.get("STAMP_DUMP"):
    np.save(os.environ["STAMP_DUMP"], s - t0)
if os.environ.get("TAIL_STAMPS"):
    a = hp
    print("TAIL (finishing wave, last segment): arrival->merged %.2f  merged->store issued %.2f  issued->acked %.2f (medians); p95 %.2f %.2f %.2f" % (
        us(np.median(a[:, 6] - a[:, 4])), us(np.median(a[:, 7] - a[:, 6])), us(np.median(a[:, 5] - a[:, 7])),
        us(np.percentile(a[:, 6] - a[:, 4], 95)), us(np.percentile(a[:, 7] - a[:, 6], 95)), us(np.percentile(a[:, 5] - a[:, 7], 95))))
    lastw = (s[:, :WS, 5].max(axis=1))
    print("  last streaming wave done -> finishing saw arrival: median %.2f p95 %.2f" % (us(np.median(hp[:, 4] - lastw)), us(np.percentile(hp[:, 4] - lastw, 95))))
    sys.exit(0)
late = hp[:, 3] > hp[:, 4]
has = hp[:, 3] > 0
print(f"hand-off complete AFTER the last arrival in {int((late & has).sum())} of {int(has.sum())} merging workgroups; "
      f"when late, by median {us(np.median((hp[late & has, 3] - hp[late & has, 4]))) if (late & has).any() else 0:.2f} us")
ok = has & ~late
if ok.any():
    print(f"tail when the hand-off was complete in time: median {us(np.median(tail[ok])):.2f} max {us(tail[ok].max()):.2f}; no hand-off at all: "
          f"{us(np.median(tail[~has])) if (~has).any() else 0:.2f}")
first_seg = (hp[:, 2] - t0) * 0.01
print("first segment complete (publishers publish then): percentiles", [round(float(np.percentile(first_seg, q)), 1) for q in (5, 25, 50, 75, 95)])
# publisher (share i-1, head h) first segment done  vs  merger (share i, head h) hand-off complete
blk = np.arange(nblocks)
share_of = blk // 8
hp_all = s[:, 8]
idx = {(int(share_of[b]), int(head[b])): b for b in blk}
gaps = []
for b in blk:
    i_, h_ = int(share_of[b]), int(head[b])
    if i_ == 0 or hp_all[b, 3] == 0:
        continue
    pb = idx[(i_ - 1, h_)]
    gaps.append(((hp_all[b, 3] - hp_all[pb, 2]) * 0.01, (hp_all[pb, 2] - t0) * 0.01, (hp_all[b, 3] - t0) * 0.01, (hp_all[b, 4] - t0) * 0.01,
                 (hp_all[pb, 7] - t0) * 0.01, (hp_all[b, 6] - t0) * 0.01))
gaps = np.array(gaps)
print("publisher first-segment-done -> merger hand-off complete: median %.2f p5 %.2f p95 %.2f" % (np.median(gaps[:, 0]), np.percentile(gaps[:, 0], 5), np.percentile(gaps[:, 0], 95)))
for r in gaps[:12]:
    print("   pub done %.1f  granules stored %.1f  merger saw them %.1f  merger complete %.1f  merger last arrival %.1f" % (r[1], r[4], r[5], r[2], r[3]))
