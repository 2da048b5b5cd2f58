"""Synthetic workloads of BASELINE.json's configs, shared by bench.py and the tools.

Everything here is device set-up plus HIP-event timing around calls into mi355_attn; nothing computes on the host.
"""
from __future__ import annotations

import os
import random
from types import SimpleNamespace
from typing import List, Optional, Sequence

import torch

import mi355_attn as m
from mi355_attn import ops

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s
MFMA_PEAK_TFLOPS = 2500.0       # MI355X_MICROARCH.md: bf16 dense


def cfg2_lengths(n_copies: int = 1) -> List[int]:
    """`random.seed(0); [randint(64, 2048) for _ in range(64)]`; further copies use seeds 1, 2, ..."""
    lens: List[int] = []
    for c in range(n_copies):
        random.seed(c)
        lens += [random.randint(64, 2048) for _ in range(64)]
    return lens


def attn_bytes_per_layer(lens: Sequence[int], Hq: int, Hkv: int, D: int, ps: int) -> int:
    """bytes_all = 4*Hkv*D*sum(L) + 4*B*Hq*D + 4*sum(ceil(L/ps)) + 16*B."""
    B = len(lens)
    return 4 * Hkv * D * sum(lens) + 4 * B * Hq * D + 4 * sum(-(-l // ps) for l in lens) + 16 * B


def kv_store_bytes_per_layer(T: int, Hkv: int, D: int) -> int:
    return 8 * T * Hkv * D + 8 * T


def decode_case(dev: str, Hq: int, Hkv: int, D: int, lens: Sequence[int], layers: int, ps: int = 16, room: int = 64,
                seed: int = 1234, device_free_list: bool = True, row_pad: Optional[int] = None) -> SimpleNamespace:
    """A batch with `lens` cached tokens per request in `layers` distinct, randomly filled KV pools, physical pages
    handed out in shuffled order, ready for prepare_for_decode (`room` further tokens per request fit)."""
    B = len(lens)
    pages = sum(-(-(l + room) // ps) for l in lens) + 8
    kv = m.KVCache(size=pages * ps, page_size=ps, head_num=Hkv, head_dim=D, dtype=torch.bfloat16, layer_num=layers,
                   device=dev, device_free_list=device_free_list, row_pad=row_pad)
    pm = m.PageManager(page_size=ps, max_req_num=B, max_context_len=(max(lens) + room) // ps * ps + 2 * ps, device=dev)
    kv.set_free_pages(torch.randperm(pages, generator=torch.Generator().manual_seed(0)).tolist())
    gg = torch.Generator(device=dev).manual_seed(seed)
    for l in range(layers):                                      # random KV content, resident in HBM
        kv._store[l].copy_(torch.randn(kv._store[l].shape, generator=gg, device=dev, dtype=torch.float32))
    backend = m.MI355AttnBackend(SimpleNamespace(device=dev, page_size=ps, kvcache=kv, page_manager=pm))
    attn = [m.Attention(i, Hq, Hkv, D) for i in range(layers)]
    reqs = [m.Req(i, list(range(l))) for i, l in enumerate(lens)]
    for r in reqs:
        r.fill_ids = r.origin_input_ids
    batch = m.Batch(reqs, pm, kv, device=dev)
    batch.prepare_for_prefill()                                  # page tables of the cached context (KV content is synthetic)
    batch.attn_backend = backend
    qkv = torch.randn((B, (Hq + 2 * Hkv) * D), generator=gg, device=dev, dtype=torch.float32).to(torch.bfloat16)
    q, k, v = qkv.split([Hq * D, Hkv * D, Hkv * D], dim=-1)      # strided views of one projection, as models/llama.py:133-134
    return SimpleNamespace(dev=dev, Hq=Hq, Hkv=Hkv, D=D, ps=ps, L=layers, B=B, kv=kv, pm=pm, backend=backend, attn=attn,
                           batch=batch, q=q, k=k, v=v, next_ids=torch.zeros(B, dtype=torch.int64, device=dev))


def decode_step(c: SimpleNamespace):
    """One decode step of the hot path: metadata (host plan + one kernel), then every layer's KV store + attention."""
    c.batch.output_ids = c.next_ids
    c.batch.prepare_for_decode()
    c.backend.init_forward_metadata(c.batch)
    o = None
    for layer in c.attn:
        o = layer(c.q, c.k, c.v, c.batch)
    return o


def decode_kernel_name(c: SimpleNamespace) -> str:
    """Which attention kernel the library runs for this batch (it decides from B and the head shape alone): the
    single-launch share kernel needs no partials workspace, the per-wave-range kernel + merge does."""
    cap = sum(c.batch.seq_lens_host)
    if ops.decode_planned_workspace_bytes(c.B, c.Hq, c.Hkv, c.D, cap, ops.pool_pitch(*c.kv.get_kv_cache(0))) == 0:
        return "m355::decode_share_kernel"
    return "m355::decode_mfma_kernel (+ m355::stream_merge_kernel)"


def time_decode_launches(c: SimpleNamespace, reps: int = 20, fused: bool = True):
    """The attention launches of the CURRENT step alone, exactly as decode_step makes them (fused KV store included),
    all layers back to back on torch's current stream, HIP events around `reps` passes.
    Returns (us per launch, algorithmic bytes per launch)."""
    md = c.backend.forward_metadata
    assert md is not None and md.decode and md.plan is not None, "run decode_step first"
    lens = list(c.batch.seq_lens_host)
    ws = c.backend._ws(ops.decode_planned_workspace_bytes(c.B, c.Hq, c.Hkv, c.D, md.plan.capacity, md.plan.pitch), c.dev)
    if ws is None:
        ws = torch.empty(16, dtype=torch.uint8, device=c.dev)
    pools = [c.kv.get_kv_cache(l) for l in range(c.L)]
    out = torch.empty((c.B, c.Hq * c.D), dtype=torch.bfloat16, device=c.dev)

    def launches():
        for kp, vp in pools:
            ops.decode_attn_planned(c.q, kp, vp, md.plan, c.Hq, workspace=ws, out=out, k_new=c.k if fused else None,
                                    v_new=c.v if fused else None)

    launches()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        launches()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (reps * c.L)
    nbytes = attn_bytes_per_layer(lens, c.Hq, c.Hkv, c.D, c.ps) + (kv_store_bytes_per_layer(c.B, c.Hkv, c.D) if fused else 0)
    return us, nbytes


def decode_shape_report(dev: str, Hq: int, Hkv: int, D: int, lens: Sequence[int], layers: int, reps: int = 20,
                        ps: int = 16, row_pad: Optional[int] = None) -> dict:
    """Launch-level roofline of decode attention for one head shape / batch (a side measurement of bench.py)."""
    c = decode_case(dev, Hq, Hkv, D, lens, layers, ps=ps, room=8, row_pad=row_pad)
    for _ in range(2):
        decode_step(c)
    us, nbytes = time_decode_launches(c, reps)
    gbps = nbytes / us / 1e3
    B = len(lens)
    return {"heads_per_gpu": [Hq, Hkv, D], "batch": B, "sum_seq_len": sum(c.batch.seq_lens_host), "layers": layers,
            "page_size": ps, "kernel": decode_kernel_name(c), "us_per_launch": round(us, 2),
            "algorithmic_bytes_per_launch": nbytes, "GBps": round(gbps, 1), "frac_of_hbm_peak": round(gbps / HBM_PEAK_GBPS, 4),
            "attn_tok_per_s_per_layer": round(B / us * 1e6, 0)}


def prefill_cfg3_case(dev: str, Hq: int = 32, Hkv: int = 8, D: int = 128, ps: int = 16, prefix: int = 1024, groups: int = 4,
                      per_group: int = 4, suffix=(128, 512), layers: int = 8) -> SimpleNamespace:
    """`groups` x `per_group` requests, every group shares one cached prefix (the SAME physical pages,
    as RadixCache hits give them), unique suffixes `random.seed(0); randint(*suffix)`; shuffled physical pages."""
    random.seed(0)
    B = groups * per_group
    ext = [random.randint(*suffix) for _ in range(B)]
    grp = [i // per_group for i in range(B)]
    exp_order = os.environ.get("M355_EXP_REQ_ORDER", "")          # experiment: batch order by new-token count (the groups' sharing is kept)
    if exp_order:
        idx = sorted(range(B), key=lambda i: ext[i], reverse=exp_order == "desc")
        ext, grp = [ext[i] for i in idx], [grp[i] for i in idx]
    pre = [prefix // ps * ps] * B
    seq = [p + e for p, e in zip(pre, ext)]
    T = sum(ext)
    npp = pre[0] // ps
    pages_total = groups * npp + sum(-(-s // ps) - npp for s in seq) + 4
    it = iter(torch.randperm(pages_total, generator=torch.Generator().manual_seed(0)).tolist())
    W = max(-(-s // ps) for s in seq) + 1
    table = torch.zeros((B, W), dtype=torch.int32)
    shared = [[next(it) for _ in range(npp)] for _ in range(groups)]
    for b in range(B):
        own = [next(it) for _ in range(-(-seq[b] // ps) - npp)]
        table[b, : npp + len(own)] = torch.tensor(shared[grp[b]] + own, dtype=torch.int32)
    g = torch.Generator(device=dev).manual_seed(0)
    pools = [(torch.randn((pages_total * ps, Hkv, D), generator=g, device=dev).to(torch.bfloat16),
              torch.randn((pages_total * ps, Hkv, D), generator=g, device=dev).to(torch.bfloat16)) for _ in range(layers)]
    q = torch.randn((T, Hq * D), generator=g, device=dev).to(torch.bfloat16)
    cu = torch.tensor([0] + list(torch.tensor(ext).cumsum(0)), dtype=torch.int32, device=dev)
    return SimpleNamespace(dev=dev, Hq=Hq, Hkv=Hkv, D=D, ps=ps, B=B, T=T, ext=ext, pre=pre, seq=seq, pools=pools, q=q, cu=cu,
                           table=table.to(dev), rows=torch.arange(B, device=dev), dseq=torch.tensor(seq, device=dev),
                           dpre=torch.tensor(pre, device=dev), out=torch.empty((T, Hq * D), dtype=torch.bfloat16, device=dev))


def time_prefill(c: SimpleNamespace, reps: int = 10) -> dict:
    """Extend attention over every pool of the case back to back, HIP events; FLOPs = 4*Hq*D*sum(E*P + E*(E+1)/2)."""
    def run():
        for kp, vp in c.pools:
            ops.prefill_attn(c.q, kp, vp, c.table, c.rows, c.dseq, c.dpre, c.cu, c.ps, c.Hq, max(c.ext), max(c.seq), out=c.out)

    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (reps * len(c.pools))
    flops = 4 * c.Hq * c.D * sum(e * p + e * (e + 1) // 2 for e, p in zip(c.ext, c.pre))
    byts = 4 * c.Hkv * c.D * sum(c.seq) + 4 * c.T * c.Hq * c.D
    return {"heads_per_gpu": [c.Hq, c.Hkv, c.D], "batch": c.B, "prefix": c.pre[0], "new_tokens": c.T, "page_size": c.ps,
            "kernel": "m355::prefill_wg_kernel", "us_per_layer": round(us, 2), "gflop_per_layer": round(flops / 1e9, 2),
            "tflops": round(flops / us / 1e6, 1), "frac_of_mfma_peak": round(flops / us / 1e6 / MFMA_PEAK_TFLOPS, 4),
            "prefill_tok_per_s_per_layer": round(c.T / us * 1e6, 0), "min_bytes_MB": round(byts / 1e6, 2)}


def cfg3_flow(dev: str, native_tree: bool = True, Hq: int = 32, Hkv: int = 8, D: int = 128, ps: int = 16, layers: int = 8,
              groups: int = 4, per_group: int = 4, prefix: int = 1024, suffix=(128, 512), decode_steps: int = 64) -> dict:
    """SURVEY §8d cfg3 END TO END, the way the scheduler drives it (engine/scheduler.py:295-297, 415-436;
    engine/batch.py:128-132): the groups' shared prefixes are already in the radix tree (one earlier request per group put
    them there), then the 16 requests are served as ONE batch:

        Req.init_next_round_input(tree)   match_prefix -> prefix_ppns            (host; native C++ tree or the Python tree)
        Batch.prepare_for_prefill()       rows, pages, positions, slots         (host plan + one HIP launch)
        per layer: KV store + prefill attention over the shared physical pages  (m355_kv_store + m355_prefill_attn)
        tree.cache_unfinished_req(req)    the new pages go into the tree        (host)
        `decode_steps` x [prepare_for_decode + per layer fused KV store + decode attention]

    Host times are wall clock around the host calls (the GPU is idle or busy elsewhere); GPU times are HIP events."""
    import time

    from mi355_attn.radix_cache import NativeRadixCache, PagedRadixCache
    random.seed(0)
    B = groups * per_group
    ext = [random.randint(*suffix) for _ in range(B)]
    rng = random.Random(1)
    prefixes = [[rng.randrange(32000) for _ in range(prefix)] for _ in range(groups)]
    prompts = [prefixes[i // per_group] + [rng.randrange(32000) for _ in range(ext[i])] for i in range(B)]
    pages = groups * (prefix // ps + 2) + sum(-(-(len(p) + decode_steps + ps) // ps) for p in prompts) + 16
    kv = m.KVCache(size=pages * ps, page_size=ps, head_num=Hkv, head_dim=D, dtype=torch.bfloat16, layer_num=layers, device=dev,
                   device_free_list=True)
    kv.set_free_pages(torch.randperm(pages, generator=torch.Generator().manual_seed(0)).tolist())
    pm = m.PageManager(page_size=ps, max_req_num=B + groups, max_context_len=prefix + suffix[1] + decode_steps + 2 * ps, device=dev)
    tree = (NativeRadixCache if native_tree else PagedRadixCache)(pm, kv)
    backend = m.MI355AttnBackend(SimpleNamespace(device=dev, page_size=ps, kvcache=kv, page_manager=pm))
    attn = [m.Attention(i, Hq, Hkv, D) for i in range(layers)]
    gg = torch.Generator(device=dev).manual_seed(7)

    def forward(batch, T):
        qkv = torch.randn((T, (Hq + 2 * Hkv) * D), generator=gg, device=dev, dtype=torch.float32).to(torch.bfloat16)
        q, k, v = qkv.split([Hq * D, Hkv * D, Hkv * D], dim=-1)
        backend.init_forward_metadata(batch)
        batch.attn_backend = backend
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for layer in attn:
            layer(q, k, v, batch)
        e1.record()
        return e0, e1

    # ---- earlier traffic: one request per group leaves its prefix in the tree (its pages stay cached when it finishes)
    for gi in range(groups):
        r = m.Req(1000 + gi, list(prefixes[gi]) + [31999], m.SamplingParams(max_new_tokens=1))
        r.init_next_round_input(tree)
        tree.inc_lock_ref(r.last_node)                       # as the scheduler does when it admits a request (scheduler.py:300-306)
        b0 = m.Batch([r], pm, kv, tree_cache=tree, device=dev)
        b0.prepare_for_prefill()
        forward(b0, len(r.fill_ids))
        r.output_ids.append(1)
        tree.cache_unfinished_req(r)                         # the life cycle of scheduler.py:415-436: cached after its prefill ...
        tree.cache_finished_req(r)                           # ... and released when it finishes (also: first-use costs land here)
    torch.cuda.synchronize()
    # ---- the 16 requests, one batch
    reqs = [m.Req(i, list(p), m.SamplingParams(max_new_tokens=decode_steps + 1)) for i, p in enumerate(prompts)]
    t0 = time.perf_counter()
    for r in reqs:
        r.init_next_round_input(tree)
        tree.inc_lock_ref(r.last_node)
    t_match = time.perf_counter() - t0
    hit = [len(r.prefix_ppns) * ps for r in reqs]
    batch = m.Batch(reqs, pm, kv, tree_cache=tree, device=dev)
    t0 = time.perf_counter()
    batch.prepare_for_prefill()
    t_prep = time.perf_counter() - t0
    T = sum(len(r.fill_ids) - h for r, h in zip(reqs, hit))
    e0, e1 = forward(batch, T)
    torch.cuda.synchronize()
    prefill_us = e0.elapsed_time(e1) * 1e3
    t0 = time.perf_counter()
    for r in reqs:
        r.output_ids.append(1)
        tree.cache_unfinished_req(r)
    t_cache = time.perf_counter() - t0
    # ---- decode steps of the same batch
    nxt = torch.zeros(B, dtype=torch.int64, device=dev)
    host_decode, ev = 0.0, []
    for _ in range(decode_steps):
        t0 = time.perf_counter()
        batch.output_ids = nxt
        batch.prepare_for_decode()
        host_decode += time.perf_counter() - t0
        ev.append(forward(batch, B))
    torch.cuda.synchronize()
    backend.check()
    dec_us = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)[len(ev) // 2]
    flops = 4 * Hq * D * sum(e * p + e * (e + 1) // 2 for e, p in zip([len(r.fill_ids) - h for r, h in zip(reqs, hit)], hit))
    return {"workload": f"{groups} groups x {per_group} requests, {prefix}-token shared prefix served from the radix tree, {suffix[0]}..{suffix[1]} new "
                        f"tokens, then {decode_steps} decode steps; {layers} layers of the Llama-3-8B attention path",
            "tree": "native (libmi355radix.so)" if native_tree else "python", "heads_per_gpu": [Hq, Hkv, D], "page_size": ps,
            "prefix_hit_tokens": hit[0] if len(set(hit)) == 1 else hit, "new_tokens": T,
            "host_match_prefix_ms_total": round(t_match * 1e3, 3), "host_prepare_for_prefill_ms": round(t_prep * 1e3, 3),
            "host_cache_unfinished_ms_total": round(t_cache * 1e3, 3),
            "prefill_us_per_layer": round(prefill_us / layers, 2), "prefill_tflops": round(flops / (prefill_us / layers) / 1e6, 1),
            "host_prepare_for_decode_us_per_step": round(host_decode / decode_steps * 1e6, 1),
            "decode_us_per_layer_median": round(dec_us / layers, 2),
            "decode_attn_path_ms_per_step": round(dec_us / 1e3, 4)}
