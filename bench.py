#!/usr/bin/env python3
"""bench.py - decode-attention hot path on MI355X (BASELINE.json configs[1]).

Workload at N=1: Llama-3.2-1B head shape (Hq=32, Hkv=8, D=64, 16 layers), batch 64 with ragged lengths
`random.seed(0); randint(64, 2048)` (sum 73 961), page_size 16, physical pages handed out in shuffled order,
distinct K/V pools per layer (2.4 GB: nothing is served from the 256 MiB Infinity Cache).

One STEP = one decode step of the hot path over the whole batch: Batch.prepare_for_decode (host plan + the
m355_decode_prepare kernel) and then, for each of the 16 layers, the KV slot write (m355_kv_store) and paged
decode attention (m355_decode_attn).  `value` = algorithmic HBM bytes of the attention path per step / step
time, summed over ranks (bytes_all = 4*Hkv*D*sum(L) + 4*B*Hq*D + 4*sum(ceil(L/ps)) + 16*B per
layer, plus the kv_store bytes).  Inputs are resident in HBM before the timed region.

N>1 (launched by torch.distributed.run, one rank per GPU): the reference's tensor parallelism - every rank
owns Hq/N query heads and Hkv/N kv heads of EVERY sequence (models/llama.py:76-87) - with the global batch
grown to 64*N so per-GPU bytes stay fixed ("weak").  The attention path has no exchange step:
the timed steps contain no collective.  The one collective of the surrounding layer - the RCCL all-reduce of the
[B_global, hidden] bf16 o_proj output (layers/linear.py:228-232) - is timed SEPARATELY after the main
measurement and reported as `allreduce_o_proj`, not folded into `value`.

`roofline`: the decode launches alone, exactly as the steps make them (decode_mfma_kernel with the fused KV
store + stream_merge_kernel; M355_BENCH_ROOFLINE_FUSED=0 times the attention-only launch instead), HIP-event
timed over all layers back to back, against 8.0 TB/s - so the rocprofv3 averages of the same command
(profiles/rNN_bench_kernel_stats.csv: main + merge) add up to `us_per_launch`.  `cpu_baseline`: the oracle (torch-native algorithm restated in
oracle/attention_ref.py) on the host cores for one layer of the same batch, rank 0, N=1 only.
"""
from __future__ import annotations

import argparse
import json
import os
import random
import sys
import time
from collections import deque
from types import SimpleNamespace

import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.abspath(__file__))
for p in (REPO, os.path.join(REPO, "mini-sglang_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MODEL = dict(name="Llama-3.2-1B", layers=16, hidden=2048, Hq=32, Hkv=8, D=64)
PAGE_SIZE = 16


def cfg2_lengths(n_copies: int):
    lens = []
    for c in range(n_copies):
        random.seed(c)              # copy 0 is exactly the BASELINE list (seed 0)
        lens += [random.randint(64, 2048) for _ in range(64)]
    return lens


def attn_bytes_per_layer(lens, Hq, Hkv, D, ps):
    B = len(lens)
    kv = 4 * Hkv * D * sum(lens)
    return kv + 4 * B * Hq * D + 4 * sum(-(-l // ps) for l in lens) + 16 * B


def kv_store_bytes_per_layer(T, Hkv, D):
    return 8 * T * Hkv * D + 8 * T


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--chunk", type=int, default=-1, help="decode chunk tokens override (-1: backend heuristic)")
    ap.add_argument("--heads", type=str, default="", help="experiment: Hq,Hkv,D per GPU instead of the 1B shape")
    ap.add_argument("--copies", type=int, default=0, help="experiment: batch = 64 * copies")
    ap.add_argument("--fixed-len", type=int, default=0, help="experiment: every request has this length (with --batch)")
    ap.add_argument("--batch", type=int, default=0, help="experiment: batch size for --fixed-len")
    ap.add_argument("--layers", type=int, default=0, help="experiment: number of layers (distinct KV pools)")
    ap.add_argument("--len-scale", type=int, default=1, help="experiment: every length x this (SURVEY 8d cfg2 'long' variant: 8)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
        args.gpus = world
    rehearse = os.environ.get("M355_BENCH_REHEARSE") == "1"            # dev: all ranks on cuda:0 over gloo (one-GPU box)
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    if world > 1:
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device(dev))     # RCCL over xGMI

    import mi355_attn as m
    from mi355_attn import ops

    from mi355_attn.tp import all_reduce_rows, row_parallel_linear, shard_heads

    tp = world
    shard = shard_heads(MODEL["Hq"], MODEL["Hkv"], MODEL["D"], tp, rank)    # refuses tp > num_kv_heads
    Hq, Hkv, D, L, hidden = shard.num_heads, shard.num_kv_heads, MODEL["D"], MODEL["layers"], MODEL["hidden"]
    if args.heads:
        Hq, Hkv, D = (int(x) for x in args.heads.split(","))
    ps = PAGE_SIZE
    lens0 = cfg2_lengths(args.copies or tp)          # global batch 64 * tp
    if args.fixed_len:
        lens0 = [args.fixed_len] * (args.batch or 1)
    if args.len_scale > 1:
        lens0 = [l * args.len_scale for l in lens0]
    if args.layers:
        L = args.layers
    B = len(lens0)
    total_steps = args.warmup + args.steps + 8
    pages_needed = sum(-(-(l + total_steps) // ps) for l in lens0) + 8
    ctx = max(2048 * args.len_scale, args.fixed_len) + 64 + total_steps

    kv = m.KVCache(size=pages_needed * ps, page_size=ps, head_num=Hkv, head_dim=D, dtype=torch.bfloat16,
                   layer_num=L, device=dev, device_free_list=os.environ.get("M355_DEVICE_FREE_LIST", "1") != "0")
    pm = m.PageManager(page_size=ps, max_req_num=B, max_context_len=ctx // ps * ps + ps, device=dev)
    g = torch.Generator().manual_seed(0)
    kv.set_free_pages(torch.randperm(pages_needed, generator=g).tolist())     # shuffled physical pages
    gg = torch.Generator(device=dev).manual_seed(1234 + rank)
    for l in range(L):                                                           # random KV content, resident in HBM
        kv._store[l].copy_(torch.randn(kv._store[l].shape, generator=gg, device=dev, dtype=torch.float32))
    backend = m.MI355AttnBackend(SimpleNamespace(device=dev, page_size=ps, kvcache=kv, page_manager=pm))
    layers = [m.Attention(i, Hq, Hkv, D) for i in range(L)]

    reqs = [m.Req(i, list(range(l))) for i, l in enumerate(lens0)]
    for r in reqs:
        r.fill_ids = r.origin_input_ids
    batch = m.Batch(reqs, pm, kv, device=dev)
    batch.prepare_for_prefill()                      # page tables for the cached context (KV content is synthetic)
    batch.attn_backend = backend
    qkv = torch.randn((B, (Hq + 2 * Hkv) * D), generator=gg, device=dev, dtype=torch.float32).to(torch.bfloat16)
    q, k, v = qkv.split([Hq * D, Hkv * D, Hkv * D], dim=-1)
    next_ids = torch.zeros(B, dtype=torch.int64, device=dev)
    red = torch.zeros((B, hidden), dtype=torch.bfloat16, device=dev)

    if args.chunk >= 0:
        import mi355_attn.backend as be
        be.pick_decode_chunk = lambda total, hkv: args.chunk

    def step():
        batch.output_ids = next_ids
        batch.prepare_for_decode()
        backend.init_forward_metadata(batch)
        for layer in layers:
            o = layer(q, k, v, batch)
        return o

    def fence():
        if tp > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    lens_start = list(batch.seq_lens_host)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if tp > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    ms_per_step = dt / args.steps * 1e3
    # bytes of step i: lengths are lens_start + i + 1
    step_bytes = 0
    for i in range(args.steps):
        cur = [l + i + 1 for l in lens_start]
        step_bytes += L * (attn_bytes_per_layer(cur, Hq, Hkv, D, ps) + kv_store_bytes_per_layer(B, Hkv, D))
    value = step_bytes * tp / dt / 1e9               # every rank moves the same number of bytes

    # ---- roofline of the dominant kernel: the steps' decode launches only, all layers back to back
    lens_now = list(batch.seq_lens_host)
    md = backend.forward_metadata
    reps = 20
    ws = backend._ws(ops.decode_workspace_bytes(B, Hq, D, md.max_seq_len, md.chunk_tokens), dev)
    pools = [kv.get_kv_cache(l) for l in range(L)]
    out = torch.empty((B, Hq * D), dtype=torch.bfloat16, device=dev)

    planned = backend.decode_mode == "planned"
    fused = planned and backend.fused_store and os.environ.get("M355_BENCH_ROOFLINE_FUSED", "1") != "0"   # the launch the steps make
    if planned:
        ws = backend._ws(ops.decode_planned_workspace_bytes(B, Hq, Hkv, D, md.plan.capacity), dev)

    def attn_only():
        for kp, vp in pools:
            if planned:
                ops.decode_attn_planned(q, kp, vp, md.plan, Hq, workspace=ws, out=out, k_new=k if fused else None,
                                        v_new=v if fused else None)
            else:
                ops.decode_attn(q, kp, vp, pm.page_table, batch.page_table_ids, batch.seq_lens, ps, Hq, md.max_seq_len,
                                chunk_tokens=md.chunk_tokens, workspace=ws, out=out)

    attn_only()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        attn_only()
    e1.record()
    torch.cuda.synchronize()
    us_per_launch = e0.elapsed_time(e1) * 1e3 / (reps * L)
    bytes_launch = attn_bytes_per_layer(lens_now, Hq, Hkv, D, ps) + (kv_store_bytes_per_layer(B, Hkv, D) if fused else 0)
    achieved = bytes_launch / (us_per_launch * 1e-6) / 1e9

    traffic = pmc_traffic(planned) if tp == 1 and not (args.heads or args.copies or args.fixed_len or args.layers or args.len_scale > 1) else None

    # ---- the layer's one collective, outside the path: o_proj's row-parallel all-reduce, timed on its own
    def time_allreduce(t, reps):
        for _ in range(3):
            all_reduce_rows(t)
        fence()
        a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a0.record()
        for _ in range(reps):
            all_reduce_rows(t)
        a1.record()
        torch.cuda.synchronize()
        us = torch.tensor([a0.elapsed_time(a1) * 1e3 / reps], device=dev, dtype=torch.float64)
        dist.all_reduce(us, op=dist.ReduceOp.MAX)
        nbytes = t.numel() * t.element_size()
        return {"tensor": f"{list(t.shape)} bf16", "bytes": nbytes, "us": round(float(us.item()), 2),
                "algbw_GBps": round(nbytes / float(us.item()) / 1e3, 1),
                "busbw_GBps": round(nbytes / float(us.item()) / 1e3 * 2 * (tp - 1) / tp, 1)}

    allreduce = allreduce_cfg4 = None
    if tp > 1:
        allreduce = time_allreduce(red, 3 if rehearse else 50)          # (gloo: ~1 s per bf16 all-reduce)
        # SURVEY 8d cfg4 (Llama-3-8B, hidden 4096): the decode-size and the prefill-size message of that model
        if not rehearse:                               # (gloo reduces 40 MB of bf16 on the host: minutes, and meaningless)
            allreduce_cfg4 = {
                "decode_B64": time_allreduce(torch.zeros((64, 4096), dtype=torch.bfloat16, device=dev), 50),
                "prefill_T4863": time_allreduce(torch.zeros((4863, 4096), dtype=torch.bfloat16, device=dev), 10)}

    # ---- f3, first measurement: o_proj (row-parallel GEMM) + its all-reduce, back to back against pipelined over
    # row blocks (mi355_attn.tp.row_parallel_linear).  Llama-3-8B shape: hidden 4096, this rank's Hq*D input columns.
    def time_o_proj(T, chunks, reps):
        if rehearse and rank == 0:
            print(f"[rehearsal] o_proj T={T} chunks={chunks}", file=sys.stderr, flush=True)
        k_in = 4096 // tp
        x = torch.randn((T, k_in), device=dev, dtype=torch.bfloat16)
        w = torch.randn((4096, k_in), device=dev, dtype=torch.bfloat16) * 0.02
        for _ in range(3):
            row_parallel_linear(x, w, chunks)
        fence()
        a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a0.record()
        for _ in range(reps):
            row_parallel_linear(x, w, chunks)
        a1.record()
        torch.cuda.synchronize()
        us = torch.tensor([a0.elapsed_time(a1) * 1e3 / reps], device=dev, dtype=torch.float64)
        dist.all_reduce(us, op=dist.ReduceOp.MAX)
        return round(float(us.item()), 2)

    o_proj = None
    if tp > 1 and (not rehearse or os.environ.get("M355_BENCH_REHEARSE_OPROJ") == "1"):
        try:
            o_proj = {"shape": f"x [T, {4096 // tp}] @ w [4096, {4096 // tp}]^T bf16, SUM over {tp} ranks", "us": {
                f"T{T}": {f"chunks{c}": time_o_proj(T, c, reps)
                          for c in cs} for T, cs, reps in (((64, (1, 2), 2), (256, (1, 4), 2)) if rehearse else
                                                           ((64, (1, 2), 50), (4863, (1, 2, 4, 8), 10)))}}
        except Exception as e:      # noqa: BLE001 - a side measurement must not take the bench line down
            o_proj = {"error": repr(e)[:200]}

    cpu = None
    if rank == 0 and tp == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(kv, pm, batch, q, lens_now, Hq, Hkv, D, ps)

    if rank == 0:
        line = {
            "metric": "decode_attn_hbm_throughput", "value": round(value, 1), "unit": "GB/s", "n_gpus": tp,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "Llama-3.2-1B decode attention path (kv_store + paged decode attn, 16 layers), "
                                   f"batch {B} ragged 64..2048 (+ctx), page_size 16, shuffled pages",
                       "global_batch": B, "sum_seq_len": sum(lens_now), "page_size": ps, "layers": L,
                       "heads_per_gpu": [Hq, Hkv, D], "parallelism": f"tp{tp}", "decode_mode": backend.decode_mode,
                       "decode_chunk_tokens": md.chunk_tokens},
            "pct_of_hbm_peak_whole_step": round(100 * value / tp / HBM_PEAK_GBPS, 2),
            "roofline": {"kernel": ("m355::decode_stream_kernel" if os.environ.get("M355_STREAM_KERNEL", "mfma")[:1] == "v"
                                     else "m355::decode_mfma_kernel") + " (+ stream_merge_kernel)" if planned
                         else "m355::decode_split_kernel (+ decode_merge_kernel)", "bound": "hbm",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                         "us_per_launch": round(us_per_launch, 3), "algorithmic_bytes_per_launch": bytes_launch,
                         "launch": "attention + fused KV store" if fused else "attention only"},
            "cpu_baseline": cpu,
        }
        if allreduce is not None:
            line["allreduce_o_proj"] = allreduce
            line["allreduce_o_proj_llama3_8b"] = allreduce_cfg4
            line["o_proj_all_reduce"] = o_proj
        print(json.dumps(line), flush=True)
    if tp > 1:
        dist.destroy_process_group()


def pmc_traffic(planned: bool):
    """HBM bytes per attention launch (main + merge kernel) from the newest committed PMC summary of this same
    command (profiles/rNN_pmc_traffic.json, written by tools/summarize_profile.py from separate FETCH_SIZE and
    WRITE_SIZE passes, already corrected per MI355X_MICROARCH.md).  None when no summary covers the kernels."""
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None
    with open(files[-1]) as f:
        kernels = json.load(f)["kernels"]
    main = "decode_stream_kernel" if os.environ.get("M355_STREAM_KERNEL", "mfma")[:1] == "v" else "decode_mfma_kernel"
    want = (main, "stream_merge_kernel") if planned else ("decode_split_kernel", "decode_merge_kernel")
    tot = 0
    for w in want:
        hit = [e["hbm_bytes_per_launch"] for k, e in kernels.items() if w in k]
        if not hit:
            return None
        tot += hit[0]
    return tot


def cpu_baseline(kv, pm, batch, q, lens, Hq, Hkv, D, ps):
    """The oracle's decode attention (per-sequence gather + SDPA, reference torch_native_backend.py:121-193) on
    the host cores: ONE layer of the same batch, 2 warm-ups + 5 timed runs (~10 s)."""
    from oracle import attention_ref as A
    n_thr = torch.get_num_threads()
    kp, vp = (t.cpu() for t in kv.get_kv_cache(0))
    table = pm.page_table.cpu()
    rows, sl = batch.page_table_ids.cpu(), batch.seq_lens.cpu()
    qc = q.cpu().reshape(len(lens), Hq, D)
    for _ in range(2):
        A.decode_attn_ref(qc, kp, vp, table, rows, sl, ps)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        A.decode_attn_ref(qc, kp, vp, table, rows, sl, ps)
        ts.append(time.perf_counter() - t0)
    ts.sort()
    med = ts[len(ts) // 2]
    return {"value": round(attn_bytes_per_layer(lens, Hq, Hkv, D, ps) / med / 1e9, 3), "unit": "GB/s", "cores": n_thr,
            "kind": "port", "sample": f"decode attention of 1 of 16 layers, same batch, bf16, median of 5 ({med * 1e3:.1f} ms/layer)"}


if __name__ == "__main__":
    main()
