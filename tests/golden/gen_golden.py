#!/usr/bin/env python3
"""Generate the golden fixtures by running the REFERENCE itself on CPU.

Run in the build container only (needs /root/reference; the GPU box has neither):

    python tests/golden/gen_golden.py

It imports the reference's torch-native attention backend, KVCache, PageManager, Batch/Req and
PagedRadixCache from /root/reference (nothing is copied), drives them with the deterministic
inputs of ``synth.py`` and writes small ``.npz`` / ``.json`` fixtures next to this file.
Shims used here only: ``sys.modules['litellm']`` stub exposing
``functools.partial`` (radix_cache.py:6 imports it from the wrong package), ``batch.device='cpu'``
(class default is 'cuda', batch.py:210), a SimpleNamespace in place of ModelRunner, and the radix
tree's ``time.time`` replaced by a strictly increasing counter so LRU order is deterministic.

While generating it also asserts that the oracle (``oracle/``) reproduces every reference result:
integers and bf16 outputs bit-exactly, fp32 outputs exactly (same torch ops, same order).
"""
from __future__ import annotations

import functools
import itertools
import json
import os
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True

_stub = types.ModuleType("litellm")
_stub.partial = functools.partial
sys.modules["litellm"] = _stub
sys.path.insert(0, "/root/reference")

from minisglang.engine.batch import Batch, Req  # noqa: E402
from minisglang.layers.attention import Attention  # noqa: E402
from minisglang.layers.attention_backends.torch_native_backend import TorchNativeAttnBackend  # noqa: E402
from minisglang.layers.sampler import SamplingParams  # noqa: E402
from minisglang.memory import radix_cache as ref_radix  # noqa: E402
from minisglang.memory.kvcache import KVCache  # noqa: E402
from minisglang.memory.page_manager import PageManager  # noqa: E402

from synth import bits16, fnv64, synth, synth_ints  # noqa: E402
from oracle import attention_ref as A  # noqa: E402
from oracle import memory_ref as M  # noqa: E402
from oracle import radix_ref as R  # noqa: E402

torch.set_num_threads(1)   # reduction order inside SDPA must not depend on the thread count


# --------------------------------------------------------------------------------------------
# flow fixtures: prefill(cut) -> extend over cached prefix -> decode steps   (bench_one_batch shape)
# --------------------------------------------------------------------------------------------
FLOW_CASES = [
    # name, ps, (Hq, Hkv, D), lens, cuts, decode_steps
    ("ps1_tiny", 1, (4, 2, 64), [9, 3, 14], [4, 1, 6], 3),
    ("ps4_tiny", 4, (4, 2, 64), [37, 5, 64], [6, 3, 23], 6),
    ("ps16_tiny", 16, (4, 2, 64), [37, 5, 64], [20, 3, 48], 18),
    ("ps256_tiny", 256, (4, 2, 64), [300, 255, 600], [270, 100, 512], 3),
    ("ps4_mha", 4, (4, 4, 64), [10, 7], [4, 2], 3),
    ("ps16_1b", 16, (32, 8, 64), [37, 5, 64], [20, 3, 48], 2),
    ("ps16_8b", 16, (32, 8, 128), [21, 24], [16, 7], 1),
    ("ps16_70b_tp8", 16, (8, 1, 128), [70, 33, 18], [64, 1, 17], 2),
    ("ps1_8b_tp2", 1, (16, 4, 128), [12, 30], [5, 11], 2),
    # BASELINE.json configs[0]: Llama-3.2-1B, page_size 1, batch 1, 128-token prompt (cut 4 like
    # bench_one_batch.py:147-188), 16 decode steps
    ("ps1_1b", 1, (32, 8, 64), [128], [4], 16),
    # IEEE half (the reference takes the dtype from the checkpoint, utils/model_config.py:40-42; Llama-2-style checkpoints
    # are fp16): Llama-2-7B at TP4 per rank, MHA.  The fixture also carries the reference's fp16 pools / outputs as hashes.
    ("ps16_llama2_f16", 16, (8, 8, 128), [70, 33, 18], [64, 1, 17], 3, "f16"),
]


def qkv_views(T, Hq, Hkv, D, seed, dtype):
    """q, k, v as strided views of one packed projection output, as models/llama.py:133-134 makes them."""
    qkv = synth((T, (Hq + 2 * Hkv) * D), seed, amp=2.0, dtype=dtype)
    q, k, v = qkv.split([Hq * D, Hkv * D, Hkv * D], dim=-1)
    return q, k, v


def run_flow(name, ps, heads, lens, cuts, steps, dtype):
    Hq, Hkv, D = heads
    n = len(lens)
    ctx = max(lens) + steps + 8
    ctx = -(-ctx // ps) * ps + ps
    S_pages = sum(-(-c // ps) for c in cuts) + sum(-(-(l + steps) // ps) for l in lens) + 4
    kv = KVCache(size=S_pages * ps, page_size=ps, head_num=Hkv, head_dim=D, dtype=dtype, layer_num=1, device="cpu")
    pm = PageManager(page_size=ps, max_req_num=2 * n + 1, max_context_len=ctx, device="cpu")
    backend = TorchNativeAttnBackend(SimpleNamespace(device="cpu", page_size=ps))
    layer = Attention(layer_id=0, num_heads=Hq, num_kv_heads=Hkv, head_dim=D, scale=D ** -0.5)
    sp = SamplingParams(temperature=0, max_new_tokens=64)

    # oracle twins
    o_pool = M.RefPagePool(S_pages, ps)
    o_pt = M.RefPageTable(ps, 2 * n + 1, ctx)
    ok_pool = torch.zeros_like(kv.k_cache[0])
    ov_pool = torch.zeros_like(kv.v_cache[0])

    ids = [synth_ints(l, 100 + i, 0, 31999) for i, l in enumerate(lens)]
    out = {"ps": ps, "heads": np.array(heads), "lens": np.array(lens), "cuts": np.array(cuts),
           "steps": steps, "ctx": ctx, "S_pages": S_pages}
    for i in range(n):
        out[f"ids{i}"] = np.array(ids[i], dtype=np.int64)

    def record(tag, batch, o_meta, o_ref, o_ora):
        for f in ("input_ids", "page_table_ids", "seq_lens", "prefix_lens", "positions", "out_cache_loc"):
            ref_v = getattr(batch, f).cpu().numpy()
            assert ref_v.dtype == np.int64, (f, ref_v.dtype)
            ora_v = getattr(o_meta, f)
            assert np.array_equal(ref_v, ora_v), (name, tag, f, ref_v, ora_v)
            out[f"{tag}.{f}"] = ref_v
        used = int(-(-int(batch.seq_lens.max()) // ps))
        rows = pm.page_table[batch.page_table_ids, :used].cpu().numpy()
        assert np.array_equal(rows, o_pt.table[o_meta.page_table_ids, :used]), (name, tag, "page rows")
        out[f"{tag}.page_rows"] = rows
        out[f"{tag}.free_head"] = np.array([int(x) for x in kv.free_pages[:4]], dtype=np.int64)
        assert [int(x) for x in kv.free_pages] == o_pool.free_pages
        assert torch.equal(kv.k_cache[0], ok_pool) and torch.equal(kv.v_cache[0], ov_pool), (name, tag, "pools")
        if dtype == torch.float16:
            assert torch.equal(o_ref, o_ora), (name, tag, "fp16 attention: oracle != reference")
            out[f"{tag}.f16_k_pool_hash"] = np.uint64(fnv64(bits16(kv.k_cache[0])))
            out[f"{tag}.f16_v_pool_hash"] = np.uint64(fnv64(bits16(kv.v_cache[0])))
            out[f"{tag}.f16_o_hash"] = np.uint64(fnv64(bits16(o_ref)))
        elif dtype == torch.bfloat16:
            assert torch.equal(o_ref, o_ora), (name, tag, "bf16 attention: oracle != reference")
            out[f"{tag}.k_pool_hash"] = np.uint64(fnv64(bits16(kv.k_cache[0])))
            out[f"{tag}.v_pool_hash"] = np.uint64(fnv64(bits16(kv.v_cache[0])))
            out[f"{tag}.o_bf16_hash"] = np.uint64(fnv64(bits16(o_ref)))
        else:
            assert torch.equal(o_ref, o_ora), (name, tag, "fp32 attention: oracle != reference")
            # keep fixtures small: full output when <= 256 KiB, else every k-th packed query row
            o_np = o_ref.numpy()
            k_ = max(1, -(-o_np.nbytes // (256 * 1024)))
            rows_ = np.arange(0, o_np.shape[0], k_, dtype=np.int64)
            out[f"{tag}.o_f32_rows"] = rows_
            out[f"{tag}.o_f32"] = o_np[rows_].copy()

    # ---- phase 1: prefill the first cut_i tokens, no prefix (bench_one_batch.py:147-173, 195-204)
    reqs = []
    for i in range(n):
        r = Req(rid=i, origin_input_ids=ids[i][:cuts[i]], sampling_params=sp)
        r.prefix_ppns = []
        r.fill_ids = list(r.origin_input_ids)
        reqs.append(r)
    o_reqs = [M.RefReq(fill_ids=list(ids[i][:cuts[i]])) for i in range(n)]
    batch = Batch(reqs=reqs, page_manager=pm, kvcache=kv)
    batch.device = "cpu"
    batch.prepare_for_prefill()
    o_meta = M.prepare_for_prefill_ref(o_reqs, o_pool, o_pt)
    T = int(batch.input_ids.numel())
    q, k, v = qkv_views(T, Hq, Hkv, D, seed=1, dtype=dtype)
    o_ref = backend.forward_extend(q, k, v, layer, batch)
    A.write_kv_ref(ok_pool, ov_pool, torch.from_numpy(o_meta.out_cache_loc), k.view(-1, Hkv, D), v.view(-1, Hkv, D))
    o_ora = A.extend_attn_ref(q.view(-1, Hq, D), ok_pool, ov_pool, torch.from_numpy(o_pt.table),
                              torch.from_numpy(o_meta.page_table_ids), torch.from_numpy(o_meta.seq_lens),
                              torch.from_numpy(o_meta.prefix_lens), ps).reshape(T, Hq * D)
    record("p1", batch, o_meta, o_ref, o_ora)

    # ---- phase 2: extend to the full prompt over the cached prefix (bench_one_batch.py:176-188)
    for i in range(n):
        reqs[i].fill_ids = list(ids[i])
        reqs[i].prefix_ppns = pm.page_table[i, : cuts[i] // ps].tolist()
        o_reqs[i] = M.RefReq(fill_ids=list(ids[i]), prefix_ppns=[int(x) for x in o_pt.table[i, : cuts[i] // ps]])
    batch = Batch(reqs=reqs, page_manager=pm, kvcache=kv)
    batch.device = "cpu"
    batch.prepare_for_prefill()
    o_meta = M.prepare_for_prefill_ref(o_reqs, o_pool, o_pt)
    T = int(batch.input_ids.numel())
    q, k, v = qkv_views(T, Hq, Hkv, D, seed=2, dtype=dtype)
    o_ref = backend.forward_extend(q, k, v, layer, batch)
    A.write_kv_ref(ok_pool, ov_pool, torch.from_numpy(o_meta.out_cache_loc), k.view(-1, Hkv, D), v.view(-1, Hkv, D))
    o_ora = A.extend_attn_ref(q.view(-1, Hq, D), ok_pool, ov_pool, torch.from_numpy(o_pt.table),
                              torch.from_numpy(o_meta.page_table_ids), torch.from_numpy(o_meta.seq_lens),
                              torch.from_numpy(o_meta.prefix_lens), ps).reshape(T, Hq * D)
    record("p2", batch, o_meta, o_ref, o_ora)

    # ---- phase 3: decode steps (bench_one_batch.py:207-214)
    for s in range(steps):
        nxt = synth_ints(n, 900 + s, 0, 31999)
        batch.output_ids = torch.tensor(nxt, dtype=torch.int64)
        batch.prepare_for_decode()
        o_meta = M.prepare_for_decode_ref(o_meta, nxt, o_pool, o_pt)
        q, k, v = qkv_views(n, Hq, Hkv, D, seed=10 + s, dtype=dtype)
        o_ref = backend.forward_decode(q, k, v, layer, batch)
        A.write_kv_ref(ok_pool, ov_pool, torch.from_numpy(o_meta.out_cache_loc), k.view(-1, Hkv, D), v.view(-1, Hkv, D))
        o_ora = A.decode_attn_ref(q.reshape(n, Hq, D), ok_pool, ov_pool, torch.from_numpy(o_pt.table),
                                  torch.from_numpy(o_meta.page_table_ids), torch.from_numpy(o_meta.seq_lens),
                                  ps).reshape(n, Hq * D)
        record(f"d{s}", batch, o_meta, o_ref, o_ora)
    return out


def gen_flows(only=None):
    for name, ps, heads, lens, cuts, steps, *flags in FLOW_CASES:
        if only and name not in only:
            continue
        a = run_flow(name, ps, heads, lens, cuts, steps, torch.bfloat16)
        b = run_flow(name, ps, heads, lens, cuts, steps, torch.float32)
        merged = dict(a)
        if "f16" in flags:                                   # the same flow once more with IEEE-half pools and inputs
            for k_, v_ in run_flow(name, ps, heads, lens, cuts, steps, torch.float16).items():
                if ".f16_" in k_:
                    merged[k_] = v_
                elif k_ in a and isinstance(v_, np.ndarray) and not k_.endswith("_hash"):
                    assert np.array_equal(a[k_], v_), (name, k_)
        for k_, v_ in b.items():
            if k_.endswith(".o_f32") or k_.endswith(".o_f32_rows"):
                merged[k_] = v_
            elif k_ in a and isinstance(v_, np.ndarray):
                assert np.array_equal(a[k_], v_), (name, k_)
        path = os.path.join(HERE, f"flow_{name}.npz")
        np.savez_compressed(path, **merged)
        print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


# --------------------------------------------------------------------------------------------
# radix traces
# --------------------------------------------------------------------------------------------
def _patch_clock():
    counter = itertools.count(1)
    ref_radix.time = SimpleNamespace(time=lambda: float(next(counter)))


def radix_trace(ps, prompts, n_pages, finish_order, evict_pages):
    """Serve `prompts` one after another the way the scheduler does (scheduler.py:295-297, 415-436):
    match_prefix via init_next_round_input -> prepare_for_prefill -> cache_unfinished_req, then one
    decode token each, then cache_finished_req in `finish_order`, then evict."""
    _patch_clock()
    ctx = (max(len(p) for p in prompts) + 8) // ps * ps + ps
    kv = KVCache(size=n_pages * ps, page_size=ps, head_num=1, head_dim=8, dtype=torch.bfloat16, layer_num=1, device="cpu")
    pm = PageManager(page_size=ps, max_req_num=len(prompts) + 1, max_context_len=ctx, device="cpu")
    tree = ref_radix.PagedRadixCache(pm, kv)
    sp = SamplingParams(temperature=0, max_new_tokens=4)

    o_pool = M.RefPagePool(n_pages, ps)
    o_pt = M.RefPageTable(ps, len(prompts) + 1, ctx)
    o_tree = R.RefRadixCache(o_pt, o_pool, ps)

    steps = []
    reqs, o_reqs, o_nodes = [], [], []

    def snap(op, extra):
        free = [int(x) for x in kv.free_pages]
        assert free == o_pool.free_pages, (op, free, o_pool.free_pages)
        assert tree.evictable_size() == o_tree.evictable and tree.protected_size() == o_tree.protected, op
        assert tree.total_size() == o_tree.total_size(), op
        d = {"op": op, "free_pages": free, "evictable": tree.evictable_size(), "protected": tree.protected_size(),
             "total": tree.total_size()}
        d.update(extra)
        steps.append(d)

    for i, p in enumerate(prompts):
        r = Req(rid=i, origin_input_ids=list(p), sampling_params=sp)
        r.init_next_round_input(tree)
        orq = M.RefReq(fill_ids=list(p))
        o_prefix, o_node = o_tree.match_prefix(orq.fill_ids[:max(len(orq.fill_ids) - 1, 0)])   # batch.py:137-150
        orq.prefix_ppns = o_prefix
        assert [int(x) for x in r.prefix_ppns] == o_prefix, (i, r.prefix_ppns, o_prefix)
        b = Batch(reqs=[r], page_manager=pm, kvcache=kv)
        b.device = "cpu"
        b.prepare_for_prefill()
        om = M.prepare_for_prefill_ref([orq], o_pool, o_pt)
        assert np.array_equal(b.out_cache_loc.numpy(), om.out_cache_loc)
        used = -(-len(p) // ps)
        snap("prefill", {"req": i, "prefix_ppns": o_prefix, "prefix_len": int(b.prefix_lens[0]),
                         "row": int(r.page_table_id), "page_row": pm.page_table[r.page_table_id, :used].tolist(),
                         "out_cache_loc": b.out_cache_loc.tolist()})
        r.output_ids.append(7000 + i)
        tree.cache_unfinished_req(r)
        o_node = o_tree.cache_unfinished_req(orq, o_node)
        assert [int(x) for x in r.prefix_ppns] == orq.prefix_ppns
        assert np.array_equal(pm.page_table.numpy(), o_pt.table)
        snap("cache_unfinished", {"req": i, "prefix_ppns": orq.prefix_ppns,
                                  "page_row": pm.page_table[r.page_table_id, :used].tolist()})
        reqs.append(r)
        o_reqs.append(orq)
        o_nodes.append(o_node)

    for i in finish_order:
        r, orq = reqs[i], o_reqs[i]
        r.output_ids.append(7100 + i)
        toks = (r.origin_input_ids + r.output_ids)[:-1]
        tree.cache_finished_req(r)
        o_tree.cache_finished_req(orq, toks, o_nodes[i])
        assert pm.free_slots == o_pt.free_rows
        snap("cache_finished", {"req": i, "free_rows": list(pm.free_slots)})

    for npg in evict_pages:
        tree.evict(npg)
        o_tree.evict(npg)
        snap("evict", {"num_pages": npg})

    probe = prompts[0] + [1, 2, 3]
    val, _ = tree.match_prefix(list(probe))
    oval, _ = o_tree.match_prefix(list(probe))
    assert [int(x) for x in val] == oval
    snap("match", {"key": list(probe), "ppns": oval})
    return {"ps": ps, "prompts": [list(p) for p in prompts], "n_pages": n_pages,
            "finish_order": list(finish_order), "evict_pages": list(evict_pages), "steps": steps}


def gen_radix():
    traces = {
        # SURVEY.md §8c example: three prompts sharing page-aligned prefixes at ps=4
        "ps4_three_prompts": radix_trace(
            4, [list(range(10)), list(range(8)) + [50, 51, 52, 53, 54], list(range(6)) + [70, 71, 72]],
            n_pages=24, finish_order=[1, 0, 2], evict_pages=[2, 50]),
        # ps=1: token-granular splits, including the repeated-pattern walk (tree "abab..", query "abc..")
        "ps1_splits": radix_trace(
            1, [[1, 2, 1, 2, 9, 9], [1, 2, 3, 4], [1, 2, 1, 2, 5], [8, 8, 8]],
            n_pages=40, finish_order=[0, 3, 1, 2], evict_pages=[3, 4, 100]),
        # ps=16 with a long shared system prompt (the generated-shared-prefix shape of test/client.sh)
        "ps16_shared_prefix": radix_trace(
            16, [list(range(48)) + [100 + j for j in range(20)],
                 list(range(48)) + [200 + j for j in range(9)],
                 list(range(32)) + [300 + j for j in range(40)]],
            n_pages=32, finish_order=[2, 0, 1], evict_pages=[1, 3, 64]),
    }
    path = os.path.join(HERE, "radix_traces.json")
    with open(path, "w") as f:
        json.dump(traces, f, indent=0, separators=(",", ":"))
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


# --------------------------------------------------------------------------------------------
# rope: reference in-tree formula (rotary_embedding.py:8-18) with its cos|sin cache (:45-63, :116-137)
# sgl_kernel itself is not importable, so only the cache construction is taken from the reference source
# by re-deriving it here; the kernel parity stays "unpinned".
# --------------------------------------------------------------------------------------------

if __name__ == "__main__":
    # no arguments: everything; otherwise only the named flow cases (the others' files are left as they are)
    only = set(sys.argv[1:])
    gen_flows(only)
    if not only:
        gen_radix()
