"""GPU parity: the HIP path (through the C ABI, via mi355_attn) against the oracle and the golden vectors.

Integer / byte results (page tables, positions, out_cache_loc, KV pools) must be bit-exact.  Attention output
is bf16 and is held to BF16_ATOL/BF16_RTOL of the fp32 reference output stored in the fixtures
(flow_util.BF16_*: 2e-2 absolute + 2e-2 relative, |v| <= 2 in these inputs)."""
import os
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from flow_util import BF16_ATOL, BF16_RTOL, FlowReplay, fixture_id, flow_fixtures
from synth import bits16, fnv64, synth
from oracle import attention_ref as A
from oracle import memory_ref as M

pytestmark = pytest.mark.gpu

DEV = "cuda"
META_FIELDS = ("input_ids", "page_table_ids", "seq_lens", "prefix_lens", "positions", "out_cache_loc")


def _mods():
    import mi355_attn
    from mi355_attn import ops
    return mi355_attn, ops


@pytest.mark.parametrize("device_free_list", [False, True], ids=["host_fifo", "device_ring"])
@pytest.mark.parametrize("path", flow_fixtures(), ids=fixture_id)
def test_flow_through_backend(path, device_free_list):
    """prefill -> extend over cached prefix -> decode steps, driven through Batch + MI355AttnBackend exactly as
    bench_one_batch drives the reference (bench_one_batch.py:195-259).  With the device free list the decode steps
    take their pages inside the metadata kernel; every page id must still be the reference's."""
    m, ops = _mods()
    fr = FlowReplay(path, torch.bfloat16)
    ps, Hq, Hkv, D = fr.ps, fr.Hq, fr.Hkv, fr.D
    kv = m.KVCache(size=fr.S_pages * ps, page_size=ps, head_num=Hkv, head_dim=D, dtype=torch.bfloat16, layer_num=1, device=DEV,
                   device_free_list=device_free_list)
    pm = m.PageManager(page_size=ps, max_req_num=fr.max_req, max_context_len=fr.ctx, device=DEV)
    backend = m.MI355AttnBackend(SimpleNamespace(device=DEV, page_size=ps, kvcache=kv, page_manager=pm))
    layer = m.Attention(layer_id=0, num_heads=Hq, num_kv_heads=Hkv, head_dim=D, scale=D ** -0.5)
    reqs, batch = None, None
    for ph in fr.phases():
        if ph.tag == "p1":
            reqs = [m.Req(i, fr.ids[i][: fr.cuts[i]]) for i in range(fr.n)]
            for r in reqs:
                r.fill_ids = list(r.origin_input_ids)
            batch = m.Batch(reqs, page_manager=pm, kvcache=kv, device=DEV)
            batch.prepare_for_prefill()
        elif ph.tag == "p2":
            for i, r in enumerate(reqs):
                r.fill_ids = list(fr.ids[i])
                r.prefix_ppns = pm.page_table[i, : fr.cuts[i] // ps].tolist()
            batch = m.Batch(reqs, page_manager=pm, kvcache=kv, device=DEV)
            batch.prepare_for_prefill()
        else:
            batch.output_ids = torch.tensor(ph.next_ids, dtype=torch.int64, device=DEV)
            batch.prepare_for_decode()
        # ---- metadata: bit-exact against the reference's own values
        for f in META_FIELDS:
            got = getattr(batch, f).cpu().numpy()
            assert got.dtype == np.int64, (ph.tag, f)
            assert np.array_equal(got, fr.golden(ph.tag, f)), (ph.tag, f, got, fr.golden(ph.tag, f))
        used = -(-int(ph.meta.seq_lens.max()) // ps)
        rows = pm.page_table[batch.page_table_ids, :used].cpu().numpy()
        assert np.array_equal(rows, fr.golden(ph.tag, "page_rows")), ph.tag
        assert kv.free_pages[:4] == [int(x) for x in fr.golden(ph.tag, "free_head")]
        if device_free_list:
            assert kv.ring.snapshot() == kv.free_pages, ph.tag       # the device FIFO is the host FIFO
        # ---- attention (writes K/V first)
        qkv = ph.qkv.to(DEV)
        q, k, v = qkv.split([Hq * D, Hkv * D, Hkv * D], dim=-1)
        backend.init_forward_metadata(batch)
        batch.attn_backend = backend
        o = layer(q, k, v, batch)
        torch.cuda.synchronize()
        assert o.shape == (q.shape[0], Hq * D) and o.dtype == torch.bfloat16
        kp, vp = kv.get_kv_cache(0)
        assert torch.equal(kp.cpu(), ph.k_pool) and torch.equal(vp.cpu(), ph.v_pool), (ph.tag, "KV pools")
        assert fnv64(bits16(kp)) == int(fr.golden(ph.tag, "k_pool_hash"))
        assert fnv64(bits16(vp)) == int(fr.golden(ph.tag, "v_pool_hash"))
        rows_, ref = fr.golden_o_f32(ph.tag)
        np.testing.assert_allclose(o.float().cpu().numpy()[rows_], ref, rtol=BF16_RTOL, atol=BF16_ATOL, err_msg=ph.tag)
        np.testing.assert_allclose(o.float().cpu().numpy(), ph.out.float().numpy(), rtol=BF16_RTOL, atol=BF16_ATOL,
                                   err_msg=ph.tag + " vs oracle bf16")


def _f16_fixtures():
    return [p for p in flow_fixtures() if "p1.f16_o_hash" in np.load(p).files]


@pytest.mark.parametrize("path", _f16_fixtures(), ids=fixture_id)
def test_flow_through_backend_in_ieee_half(path):
    """An oracle-pinned flow in fp16 (the reference takes the dtype from the checkpoint, utils/model_config.py:40-42):
    prefill -> extend over the cached prefix -> decode steps through Batch + MI355AttnBackend with IEEE-half pools and
    inputs.  Pools bit-exact against the REFERENCE's fp16 pools (fixture hashes), metadata bit-exact, attention within
    tolerance of the reference's fp32 anchor and of the oracle's fp16 output."""
    m, ops = _mods()
    fr = FlowReplay(path, torch.float16)
    ps, Hq, Hkv, D = fr.ps, fr.Hq, fr.Hkv, fr.D
    kv = m.KVCache(size=fr.S_pages * ps, page_size=ps, head_num=Hkv, head_dim=D, dtype=torch.float16, layer_num=1, device=DEV,
                   device_free_list=True)
    pm = m.PageManager(page_size=ps, max_req_num=fr.max_req, max_context_len=fr.ctx, device=DEV)
    backend = m.MI355AttnBackend(SimpleNamespace(device=DEV, page_size=ps, kvcache=kv, page_manager=pm))
    layer = m.Attention(layer_id=0, num_heads=Hq, num_kv_heads=Hkv, head_dim=D, scale=D ** -0.5)
    reqs, batch = None, None
    for ph in fr.phases():
        if ph.tag == "p1":
            reqs = [m.Req(i, fr.ids[i][: fr.cuts[i]]) for i in range(fr.n)]
            for r in reqs:
                r.fill_ids = list(r.origin_input_ids)
            batch = m.Batch(reqs, page_manager=pm, kvcache=kv, device=DEV)
            batch.prepare_for_prefill()
        elif ph.tag == "p2":
            for i, r in enumerate(reqs):
                r.fill_ids = list(fr.ids[i])
                r.prefix_ppns = pm.page_table[i, : fr.cuts[i] // ps].tolist()
            batch = m.Batch(reqs, page_manager=pm, kvcache=kv, device=DEV)
            batch.prepare_for_prefill()
        else:
            batch.output_ids = torch.tensor(ph.next_ids, dtype=torch.int64, device=DEV)
            batch.prepare_for_decode()
        for f in META_FIELDS:
            assert np.array_equal(getattr(batch, f).cpu().numpy(), fr.golden(ph.tag, f)), (ph.tag, f)
        qkv = ph.qkv.to(DEV)
        q, k, v = qkv.split([Hq * D, Hkv * D, Hkv * D], dim=-1)
        backend.init_forward_metadata(batch)
        batch.attn_backend = backend
        o = layer(q, k, v, batch)
        torch.cuda.synchronize()
        backend.check()
        assert o.shape == (q.shape[0], Hq * D) and o.dtype == torch.float16
        kp, vp = kv.get_kv_cache(0)
        assert fnv64(bits16(kp)) == int(fr.golden(ph.tag, "f16_k_pool_hash")), (ph.tag, "K pool vs the reference's")
        assert fnv64(bits16(vp)) == int(fr.golden(ph.tag, "f16_v_pool_hash")), (ph.tag, "V pool vs the reference's")
        rows_, ref = fr.golden_o_f32(ph.tag)
        np.testing.assert_allclose(o.float().cpu().numpy()[rows_], ref, rtol=BF16_RTOL, atol=BF16_ATOL, err_msg=ph.tag)
        np.testing.assert_allclose(o.float().cpu().numpy(), ph.out.float().numpy(), rtol=BF16_RTOL, atol=BF16_ATOL,
                                   err_msg=ph.tag + " vs oracle fp16")


def test_device_page_ring_follows_the_host_fifo():
    """Random serving trace - prefills, decode steps over page boundaries, finished requests giving pages back in
    odd orders, wrap-around of the ring - with the device free list against the same trace on the host FIFO:
    identical page tables, slots and free lists at every step (memory/kvcache.py:38-67 order)."""
    m, ops = _mods()
    rng = np.random.default_rng(11)
    for ps in (1, 4, 16):
        pages = 96
        kvs = [m.KVCache(size=pages * ps, page_size=ps, head_num=1, head_dim=64, dtype=torch.bfloat16, layer_num=1, device=DEV,
                         device_free_list=flag) for flag in (False, True)]
        pms = [m.PageManager(page_size=ps, max_req_num=16, max_context_len=64 * ps, device=DEV) for _ in range(2)]
        live = [[], []]                                              # per variant: list of (req, batch-of-one bookkeeping)
        batches = [None, None]
        rid = 0
        for step in range(60):
            action = rng.integers(0, 4)
            if action == 0 or batches[0] is None or not batches[0].reqs:
                n_new = int(rng.integers(1, 4))
                lens = [int(x) for x in rng.integers(1, 3 * ps + 2, size=n_new)]
                if kvs[0].available_page_num() < sum(-(-l // ps) for l in lens) + 8 or len(batches[0].reqs if batches[0] else []) + n_new > 12:
                    continue
                for v in range(2):
                    reqs = [m.Req(rid + i, list(range(l))) for i, l in enumerate(lens)]
                    for r in reqs:
                        r.fill_ids = list(r.origin_input_ids)
                    nb = m.Batch(reqs, page_manager=pms[v], kvcache=kvs[v], device=DEV)
                    nb.prepare_for_prefill()
                    if batches[v] is None or not batches[v].reqs:
                        batches[v] = nb
                    else:
                        batches[v].merge_batch(nb)
                rid += n_new
            elif action == 1 and len(batches[0].reqs) > 1:
                # a request finishes: its pages go back (reference scheduler: free the row's pages, then the row)
                j = int(rng.integers(0, len(batches[0].reqs)))
                for v in range(2):
                    b = batches[v]
                    row = int(b.page_table_ids[j])
                    n_pg = -(-b._lens_host()[j] // ps)
                    kvs[v]._free_pages(pms[v].page_table[row, :n_pg])
                    pms[v].free([row])
                    b.input_ids = b.output_ids = None               # (stale after merge_batch; not part of this test)
                    b.filter_batch([i for i in range(len(b.reqs)) if i != j])
            else:
                if kvs[0].available_page_num() < len(batches[0].reqs):
                    continue
                for v in range(2):
                    b = batches[v]
                    b.output_ids = torch.zeros(len(b.reqs), dtype=torch.int64, device=DEV)
                    b.prepare_for_decode()
            a, b = batches
            if a is None:
                continue
            assert torch.equal(a.seq_lens, b.seq_lens) and torch.equal(a.page_table_ids, b.page_table_ids)
            if a.out_cache_loc is not None and b.out_cache_loc is not None:
                assert torch.equal(a.out_cache_loc, b.out_cache_loc), (ps, step)
                assert torch.equal(a.positions, b.positions)
            used = -(-max(a._lens_host()) // ps)
            assert torch.equal(pms[0].page_table[a.page_table_ids, :used], pms[1].page_table[b.page_table_ids, :used]), (ps, step)
            assert kvs[0].free_pages == kvs[1].free_pages == kvs[1].ring.snapshot(), (ps, step)


def test_kv_store_edges():
    m, ops = _mods()
    S, Hkv, D = 64, 2, 64
    kp = torch.zeros((S, Hkv, D), dtype=torch.bfloat16, device=DEV)
    vp = torch.zeros_like(kp)
    # empty write
    e = torch.empty((0, Hkv * D), dtype=torch.bfloat16, device=DEV)
    ops.kv_store(kp, vp, torch.empty(0, dtype=torch.int64, device=DEV), e, e)
    # strided sources, unordered slots, a negative slot (torch wraps it), last slot
    T = 5
    qkv = synth((T, 6 * D), 7, amp=2.0).to(DEV)
    k, v = qkv[:, 2 * D:4 * D], qkv[:, 4 * D:]
    loc = torch.tensor([63, 0, 17, -2, 5], dtype=torch.int64, device=DEV)
    ops.kv_store(kp, vp, loc, k.view(T, Hkv, D), v.view(T, Hkv, D))
    rk, rv = torch.zeros((S, Hkv, D), dtype=torch.bfloat16), torch.zeros((S, Hkv, D), dtype=torch.bfloat16)
    A.write_kv_ref(rk, rv, loc.cpu(), k.cpu().view(T, Hkv, D), v.cpu().view(T, Hkv, D))
    assert torch.equal(kp.cpu(), rk) and torch.equal(vp.cpu(), rv)
    # fp16 and fp32 pools copy bytes just the same
    for dt in (torch.float16, torch.float32):
        kp2 = torch.zeros((S, Hkv, D), dtype=dt, device=DEV)
        vp2 = torch.zeros_like(kp2)
        src = synth((T, 2 * Hkv * D), 9, dtype=dt).to(DEV)
        ops.kv_store(kp2, vp2, loc, src[:, : Hkv * D], src[:, Hkv * D:])
        rk2, rv2 = torch.zeros((S, Hkv, D), dtype=dt), torch.zeros((S, Hkv, D), dtype=dt)
        A.write_kv_ref(rk2, rv2, loc.cpu(), src.cpu()[:, : Hkv * D].view(T, Hkv, D), src.cpu()[:, Hkv * D:].view(T, Hkv, D))
        assert torch.equal(kp2.cpu(), rk2) and torch.equal(vp2.cpu(), rv2)
    # CPU tensors are refused loudly (no fallback)
    with pytest.raises(RuntimeError):
        ops.kv_store(kp.cpu(), vp.cpu(), loc.cpu(), k.cpu(), v.cpu())


def test_out_cache_loc_kernel():
    m, ops = _mods()
    rng = np.random.default_rng(0)
    for ps in (1, 3, 16, 256):
        table = rng.integers(0, 1 << 20, size=(9, 40), dtype=np.int32)
        rows = rng.integers(0, 9, size=777).astype(np.int64)
        pos = rng.integers(0, 40 * ps, size=777).astype(np.int64)
        got = ops.out_cache_loc(torch.from_numpy(table).to(DEV), torch.from_numpy(rows).to(DEV),
                                torch.from_numpy(pos).to(DEV), ps)
        assert got.dtype == torch.int64
        assert np.array_equal(got.cpu().numpy(), M.out_cache_loc_ref(table, rows, pos, ps))


def _random_paged_case(B, Hq, Hkv, D, ps, lens, seed, shuffle=True):
    """Pools + page table with physical pages handed out in a shuffled order (non-contiguous gathers)."""
    g = torch.Generator().manual_seed(seed)
    npages = sum(-(-l // ps) for l in lens) + 3
    perm = torch.randperm(npages, generator=g).tolist() if shuffle else list(range(npages))
    W = max(-(-l // ps) for l in lens) + 2
    table = np.full((B + 2, W), 0, dtype=np.int32)
    rows = list(range(B, 0, -1))           # rows deliberately not 0..B-1
    it = iter(perm)
    for b, l in enumerate(lens):
        for j in range(-(-l // ps)):
            table[rows[b], j] = next(it)
        table[rows[b], -(-l // ps):] = npages              # stale entries past the live length -> poison page
    kp = (torch.randn(((npages + 1) * ps, Hkv, D), generator=g) * 1.0).to(torch.bfloat16)
    vp = (torch.randn(((npages + 1) * ps, Hkv, D), generator=g) * 1.0).to(torch.bfloat16)
    kp[npages * ps:] = float("nan")                        # in-bounds, but any use of it shows up as NaN
    vp[npages * ps:] = float("nan")
    return table, np.asarray(rows, dtype=np.int64), kp, vp


@pytest.mark.parametrize("Hq,Hkv,D", [(32, 8, 64), (32, 8, 128), (8, 1, 128), (16, 4, 128), (4, 4, 64), (4, 2, 128)])
@pytest.mark.parametrize("ps", [1, 16])
def test_decode_attn_split_and_merge(Hq, Hkv, D, ps):
    """Ragged lengths around every chunk boundary, every chunk size; oracle = fp64 formula."""
    m, ops = _mods()
    lens = [1, 2, 63, 64, 65, 127, 129, 300, 777, 1025]
    B = len(lens)
    table, rows, kp, vp = _random_paged_case(B, Hq, Hkv, D, ps, lens, seed=Hq + D + ps)
    g = torch.Generator().manual_seed(1)
    q = (torch.randn((B, Hq * D), generator=g) * 2).to(torch.bfloat16)
    ref = A.paged_attn_f64(q.view(B, Hq, D), kp, vp, torch.from_numpy(table), torch.from_numpy(rows),
                           torch.tensor(lens), None, ps).reshape(B, Hq * D).float().numpy()
    dq, dk, dv = q.to(DEV), kp.to(DEV), vp.to(DEV)
    dt, dr, dl = torch.from_numpy(table).to(DEV), torch.from_numpy(rows).to(DEV), torch.tensor(lens, device=DEV)
    for hint in (max(lens), max(lens) + 500):         # the self-contained call: its own plan, laid out for B * hint tokens
        o = ops.decode_attn(dq, dk, dv, dt, dr, dl, ps, Hq, hint)
        torch.cuda.synchronize()
        np.testing.assert_allclose(o.float().cpu().numpy(), ref, rtol=BF16_RTOL, atol=BF16_ATOL, err_msg=f"hint={hint}")
    for cap in (sum(lens), sum(lens) + 1000):
        plan = ops.decode_plan(dt, dr, dl, ps, Hq, Hkv, D, cap)
        o = ops.decode_attn_planned(dq, dk, dv, plan, Hq)
        torch.cuda.synchronize()
        np.testing.assert_allclose(o.float().cpu().numpy(), ref, rtol=BF16_RTOL, atol=BF16_ATOL, err_msg=f"planned cap={cap}")


def _padded(pool, pad=64):
    """The same [S, Hkv, D] pool with `pad` unused elements behind every token row (mi355_attn.KVCache's layout: the
    share kernel's home; contiguous power-of-two rows at head_dim 64 go to the range kernel, csrc/decode_stream.hip)."""
    S, Hkv, D = pool.shape
    full = torch.zeros((S, Hkv * D + pad), dtype=pool.dtype, device=pool.device)
    full[:, :Hkv * D] = pool.reshape(S, Hkv * D)
    return full[:, :Hkv * D].unflatten(1, (Hkv, D))


def _gpu_decode_reference(q, kp, vp, table, rows, lens, ps, Hq):
    """torch fp32 on the GPU, one request at a time, same gather as the reference backend."""
    B = len(lens)
    Hkv, D = kp.shape[1], kp.shape[2]
    ref = torch.empty((B, Hq, D), device=q.device)
    for b, l in enumerate(lens):
        pos = torch.arange(l, device=q.device)
        slots = table[rows[b], pos // ps].long() * ps + pos % ps
        k = kp[slots].float().repeat_interleave(Hq // Hkv, dim=1)
        v = vp[slots].float().repeat_interleave(Hq // Hkv, dim=1)
        s = torch.einsum("hd,lhd->hl", q[b].view(Hq, D).float(), k) * D ** -0.5
        ref[b] = torch.einsum("hl,lhd->hd", torch.softmax(s, dim=-1), v)
    return ref.view(B, -1)


def test_decode_planned_many_tiny_requests():
    """More requests inside one range than a pass stages (kMaxPieces): ranges are walked in several passes."""
    m, ops = _mods()
    Hq, Hkv, D, ps = 8, 2, 64, 4
    rng = np.random.default_rng(3)
    lens = [int(x) for x in rng.integers(1, 6, size=700)]
    B = len(lens)
    table, rows, kp, vp = _random_paged_case(B, Hq, Hkv, D, ps, lens, seed=11)
    q = (torch.randn((B, Hq * D), generator=torch.Generator().manual_seed(4)) * 2).to(torch.bfloat16)
    dq, dk, dv = q.to(DEV), kp.to(DEV), vp.to(DEV)
    dt, dr, dl = torch.from_numpy(table).to(DEV), torch.from_numpy(rows).to(DEV), torch.tensor(lens, device=DEV)
    ref = _gpu_decode_reference(dq, dk, dv, dt, dr, lens, ps, Hq)
    plan = ops.decode_plan(dt, dr, dl, ps, Hq, Hkv, D, sum(lens))
    o = ops.decode_attn_planned(dq, dk, dv, plan, Hq)
    assert torch.allclose(o.float(), ref, rtol=BF16_RTOL, atol=BF16_ATOL)
    o2 = ops.decode_attn(dq, dk, dv, dt, dr, dl, ps, Hq, max(lens))
    assert torch.allclose(o2.float(), ref, rtol=BF16_RTOL, atol=BF16_ATOL)


@pytest.mark.parametrize("Hq,Hkv,D,ps,lens", [
    (32, 8, 64, 16, [1, 2, 17, 33, 64, 65, 300, 777, 1025, 290, 291]),
    (32, 8, 128, 16, [5, 130, 2047, 1]),
    (8, 1, 128, 1, [1, 1, 3, 900]),
    (8, 2, 64, 4, [int(x) for x in np.random.default_rng(5).integers(1, 6, size=300)]),     # many pieces per range
])
def test_decode_fused_store_equals_store_then_attention(Hq, Hkv, D, ps, lens):
    """m355_decode_store_attn_planned == m355_kv_store + m355_decode_attn_planned: pools bit-exact, output bit-exact
    (same kernel, the newest row only comes from a different address)."""
    m, ops = _mods()
    B = len(lens)
    table, rows, kp, vp = _random_paged_case(B, Hq, Hkv, D, ps, lens, seed=B + D)
    g = torch.Generator().manual_seed(9)
    qkv = (torch.randn((B, (Hq + 2 * Hkv) * D), generator=g) * 2).to(torch.bfloat16).to(DEV)
    q, k, v = qkv.split([Hq * D, Hkv * D, Hkv * D], dim=-1)                              # strided views
    dt, dr, dl = torch.from_numpy(table).to(DEV), torch.from_numpy(rows).to(DEV), torch.tensor(lens, device=DEV)
    pos = dl - 1
    loc = ops.out_cache_loc(dt, dr, pos, ps)
    plan = ops.decode_plan(dt, dr, dl, ps, Hq, Hkv, D, sum(lens))
    kp1, vp1 = kp.to(DEV), vp.to(DEV)
    ops.kv_store(kp1, vp1, loc, k, v)
    o1 = ops.decode_attn_planned(q, kp1, vp1, plan, Hq)
    kp2, vp2 = kp.to(DEV), vp.to(DEV)
    o2 = ops.decode_attn_planned(q, kp2, vp2, plan, Hq, k_new=k, v_new=v)
    torch.cuda.synchronize()
    live = torch.ones(kp1.shape[0], dtype=torch.bool, device=DEV)
    live[-ps:] = False                                                                      # the NaN poison page
    assert torch.equal(kp1[live], kp2[live]) and torch.equal(vp1[live], vp2[live])
    assert torch.equal(o1, o2)
    ref = _gpu_decode_reference(q.contiguous(), kp1, vp1, dt, rows.tolist(), lens, ps, Hq)
    assert torch.allclose(o2.float(), ref, rtol=BF16_RTOL, atol=BF16_ATOL)
    assert plan.error() == 0


def test_decode_planned_more_ranges_than_waves():
    """A token stream longer than waves x 1024 tokens: every wave walks several ranges."""
    m, ops = _mods()
    Hq, Hkv, D, ps = 8, 8, 64, 16
    lens = [70001, 64000, 59999, 61234, 66000, 63000, 58000, 65536]       # 507 770 tokens > 384 * 1024
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
    kp = _padded(torch.randn((npages * ps, Hkv, D), generator=g, device=DEV).to(torch.bfloat16))
    vp = _padded(torch.randn((npages * ps, Hkv, D), generator=g, device=DEV).to(torch.bfloat16))
    pitch = ops.pool_pitch(kp, vp)
    q = (torch.randn((B, Hq * D), generator=g, device=DEV) * 3).to(torch.bfloat16)
    dt, dr, dl = table.to(DEV), torch.arange(B, device=DEV), torch.tensor(lens, device=DEV)
    ref = _gpu_decode_reference(q, kp, vp, dt, list(range(B)), lens, ps, Hq)
    plan = ops.decode_plan(dt, dr, dl, ps, Hq, Hkv, D, sum(lens), pool_row_stride=pitch)
    o = ops.decode_attn_planned(q, kp, vp, plan, Hq)
    assert torch.allclose(o.float(), ref, rtol=BF16_RTOL, atol=BF16_ATOL)
    assert ops.decode_planned_workspace_bytes(B, Hq, Hkv, D, sum(lens), pitch) == 0   # the share kernel, 3-4 publishers per merger
    assert plan.error() == 0
    # the same batch on contiguous pools (the reference's layout): several ranges per wave of the range kernel
    kc, vc = kp.contiguous(), vp.contiguous()
    assert ops.decode_planned_workspace_bytes(B, Hq, Hkv, D, sum(lens), 0) > 0
    plan_c = ops.decode_plan(dt, dr, dl, ps, Hq, Hkv, D, sum(lens))
    assert torch.allclose(ops.decode_attn_planned(q, kc, vc, plan_c, Hq).float(), ref, rtol=BF16_RTOL, atol=BF16_ATOL)
    with pytest.raises(Exception, match="pitch"):                                     # a plan is made for ONE pool layout
        ops.decode_attn_planned(q, kp, vp, plan_c, Hq)


def test_decode_attn_softmax_spike():
    """One key dominates (forces the running-max rescale late in a chunk and across chunks)."""
    m, ops = _mods()
    Hq, Hkv, D, ps = 8, 2, 64, 16
    lens = [500]
    table, rows, kp, vp = _random_paged_case(1, Hq, Hkv, D, ps, lens, seed=5)
    q = torch.ones((1, Hq * D), dtype=torch.bfloat16)
    slot = int(table[rows[0], 431 // ps]) * ps + 431 % ps
    kp[slot] = 4.0                                      # q.k = 4*64 -> towers over N(0, 8) scores
    ref = A.paged_attn_f64(q.view(1, Hq, D), kp, vp, torch.from_numpy(table), torch.from_numpy(rows),
                           torch.tensor(lens), None, ps).reshape(1, -1).float().numpy()
    o = ops.decode_attn(q.to(DEV), kp.to(DEV), vp.to(DEV), torch.from_numpy(table).to(DEV),
                        torch.from_numpy(rows).to(DEV), torch.tensor(lens, device=DEV), ps, Hq, 500)
    np.testing.assert_allclose(o.float().cpu().numpy(), ref, rtol=BF16_RTOL, atol=BF16_ATOL)
    plan = ops.decode_plan(torch.from_numpy(table).to(DEV), torch.from_numpy(rows).to(DEV), torch.tensor(lens, device=DEV),
                           ps, Hq, Hkv, D, 500)
    o = ops.decode_attn_planned(q.to(DEV), kp.to(DEV), vp.to(DEV), plan, Hq)
    np.testing.assert_allclose(o.float().cpu().numpy(), ref, rtol=BF16_RTOL, atol=BF16_ATOL)


@pytest.mark.parametrize("Hq,Hkv,D,ps", [(32, 8, 64, 16), (32, 8, 128, 16), (8, 1, 128, 1), (16, 4, 128, 256), (8, 2, 128, 12), (4, 2, 64, 3)])
def test_prefill_attn_ragged_prefix(Hq, Hkv, D, ps):
    """Mixed zero / non-zero / non-page-aligned-length prefixes, extends of 1..300 tokens."""
    m, ops = _mods()
    seq = [5, 40, 333, 64, 257, 1200]
    pre = [0, 32, 48, 63, 256, 1100]
    pre = [p // ps * ps for p in pre]                   # prefixes are whole pages (batch.py:256)
    B = len(seq)
    table, rows, kp, vp = _random_paged_case(B, Hq, Hkv, D, ps, seq, seed=D + ps)
    ext = [s - p for s, p in zip(seq, pre)]
    T = sum(ext)
    g = torch.Generator().manual_seed(2)
    q = (torch.randn((T, Hq * D), generator=g) * 2).to(torch.bfloat16)
    ref = A.paged_attn_f64(q.view(T, Hq, D), kp, vp, torch.from_numpy(table), torch.from_numpy(rows),
                           torch.tensor(seq), torch.tensor(pre), ps).reshape(T, Hq * D).float().numpy()
    cu = torch.tensor(np.concatenate([[0], np.cumsum(ext)]), dtype=torch.int32, device=DEV)
    o = ops.prefill_attn(q.to(DEV), kp.to(DEV), vp.to(DEV), torch.from_numpy(table).to(DEV),
                         torch.from_numpy(rows).to(DEV), torch.tensor(seq, device=DEV), torch.tensor(pre, device=DEV),
                         cu, ps, Hq, max(ext), max(seq))
    torch.cuda.synchronize()
    np.testing.assert_allclose(o.float().cpu().numpy(), ref, rtol=BF16_RTOL, atol=BF16_ATOL)


@pytest.mark.parametrize("Hq,Hkv,D", [(32, 8, 64), (8, 1, 128)])
def test_degenerate_batches(Hq, Hkv, D):
    """Requests with nothing to do next to ordinary ones, and empty batches: an extend of zero tokens (its prefix covers the
    whole sequence) and a decode request of length zero add no rows / leave their output row alone, and the others come
    out as if the degenerate ones were not there; B == 0 and T == 0 calls return at once."""
    m, ops = _mods()
    ps = 16
    # ---- prefill: extends of 0, 7, 0, 40, 0 tokens
    seq = [32, 39, 16, 104, 0]
    pre = [32, 32, 16, 64, 0]
    B = len(seq)
    table, rows, kp, vp = _random_paged_case(B, Hq, Hkv, D, ps, [max(s, 1) for s in seq], seed=77)
    ext = [s_ - p_ for s_, p_ in zip(seq, pre)]
    T = sum(ext)
    g = torch.Generator().manual_seed(3)
    q = (torch.randn((T, Hq * D), generator=g) * 2).to(torch.bfloat16)
    live = [b for b in range(B) if ext[b] > 0]
    ref = A.paged_attn_f64(q.view(T, Hq, D), kp, vp, torch.from_numpy(table), torch.from_numpy(rows[live]),
                           torch.tensor([seq[b] for b in live]), torch.tensor([pre[b] for b in live]), ps).reshape(T, Hq * D)
    cu = torch.tensor(np.concatenate([[0], np.cumsum(ext)]), dtype=torch.int32, device=DEV)
    dk, dv, dt, dr = kp.to(DEV), vp.to(DEV), torch.from_numpy(table).to(DEV), torch.from_numpy(rows).to(DEV)
    o = ops.prefill_attn(q.to(DEV), dk, dv, dt, dr, torch.tensor(seq, device=DEV), torch.tensor(pre, device=DEV), cu, ps, Hq,
                         max(ext), max(seq))
    torch.cuda.synchronize()
    np.testing.assert_allclose(o.float().cpu().numpy(), ref.float().numpy(), rtol=BF16_RTOL, atol=BF16_ATOL)
    # ---- decode: lengths 0, 1, 17, 0, 300 (planned, with both plan capacities; the self-contained call)
    lens = [0, 1, 17, 0, 300]
    table, rows, kp, vp = _random_paged_case(B, Hq, Hkv, D, ps, [max(l, 1) for l in lens], seed=78)
    dk, dv, dt, dr = kp.to(DEV), vp.to(DEV), torch.from_numpy(table).to(DEV), torch.from_numpy(rows).to(DEV)
    qd = (torch.randn((B, Hq * D), generator=g) * 2).to(torch.bfloat16)
    livd = [b for b in range(B) if lens[b] > 0]
    refd = A.paged_attn_f64(qd.view(B, Hq, D)[livd], kp, vp, torch.from_numpy(table), torch.from_numpy(rows[livd]),
                            torch.tensor([lens[b] for b in livd]), None, ps).reshape(len(livd), Hq * D).float().numpy()
    dl = torch.tensor(lens, device=DEV)
    for cap in (sum(lens), 5000):
        plan = ops.decode_plan(dt, dr, dl, ps, Hq, Hkv, D, cap)
        out = torch.full((B, Hq * D), 7.0, dtype=torch.bfloat16, device=DEV)
        ops.decode_attn_planned(qd.to(DEV), dk, dv, plan, Hq, out=out)
        torch.cuda.synchronize()
        np.testing.assert_allclose(out[livd].float().cpu().numpy(), refd, rtol=BF16_RTOL, atol=BF16_ATOL, err_msg=f"cap {cap}")
        assert torch.isfinite(out.float()).all()
    o2 = ops.decode_attn(qd.to(DEV), dk, dv, dt, dr, dl, ps, Hq, max(lens))
    torch.cuda.synchronize()
    np.testing.assert_allclose(o2[livd].float().cpu().numpy(), refd, rtol=BF16_RTOL, atol=BF16_ATOL)
    # ---- nothing at all
    e64, e32 = torch.empty(0, dtype=torch.int64, device=DEV), torch.zeros(1, dtype=torch.int32, device=DEV)
    assert ops.prefill_attn(q[:0].to(DEV), dk, dv, dt, e64, e64, e64, e32, ps, Hq, 0, 0).shape == (0, Hq * D)
    assert ops.decode_attn(qd[:0].to(DEV), dk, dv, dt, e64, e64, ps, Hq, 16).shape == (0, Hq * D)


def _gpu_prefill_reference(q, kp, vp, table, rows, seq, pre, ps, Hq):
    """torch fp32 on the GPU, one request at a time: row j of the extend sees cached tokens 0 .. pre + j."""
    Hkv, D = kp.shape[1], kp.shape[2]
    outs, start = [], 0
    for b, (L, P) in enumerate(zip(seq, pre)):
        E = L - P
        pos = torch.arange(L, device=q.device)
        slots = table[rows[b], pos // ps].long() * ps + pos % ps
        k = kp[slots].float().repeat_interleave(Hq // Hkv, dim=1)
        v = vp[slots].float().repeat_interleave(Hq // Hkv, dim=1)
        qq = q[start:start + E].view(E, Hq, D).float()
        s = torch.einsum("ehd,lhd->hel", qq, k) * D ** -0.5
        mask = pos[None, :] > (P + torch.arange(E, device=q.device))[:, None]
        s = s.masked_fill(mask[None], float("-inf"))
        outs.append(torch.einsum("hel,lhd->ehd", torch.softmax(s, dim=-1), v).reshape(E, Hq * D))
        start += E
    return torch.cat(outs)


@pytest.mark.parametrize("Hq,Hkv,D,ps", [(8, 2, 128, 16), (8, 2, 64, 4), (8, 1, 128, 16), (4, 4, 64, 16)])
def test_prefill_long_sequences_and_late_spike(Hq, Hkv, D, ps):
    """Several slot windows (> 1024 cached tokens), extends longer than a workgroup's row block, odd tile counts, and
    one key that dominates late (the lazy running max must still rescale)."""
    m, ops = _mods()
    seq = [2500, 2311, 1057, 33]
    pre = [0, 2096, 1024, 32]
    pre = [p // ps * ps for p in pre]
    B = len(seq)
    table, rows, kp, vp = _random_paged_case(B, Hq, Hkv, D, ps, seq, seed=3 * D + ps)
    slot = int(table[rows[0], 2200 // ps]) * ps + 2200 % ps
    kp[slot] = 1.5                                       # scores of ~ +1.5 * sum(q) tower over N(0, D) for most rows
    ext = [s - p for s, p in zip(seq, pre)]
    T = sum(ext)
    g = torch.Generator().manual_seed(12)
    q = (torch.randn((T, Hq * D), generator=g) + 0.5).to(torch.bfloat16).to(DEV)
    dk, dv = kp.to(DEV), vp.to(DEV)
    dt, dr = torch.from_numpy(table).to(DEV), torch.from_numpy(rows).to(DEV)
    ref = _gpu_prefill_reference(q, dk, dv, dt, rows.tolist(), seq, pre, ps, Hq)
    cu = torch.tensor(np.concatenate([[0], np.cumsum(ext)]), dtype=torch.int32, device=DEV)
    o = ops.prefill_attn(q, dk, dv, dt, dr, torch.tensor(seq, device=DEV), torch.tensor(pre, device=DEV), cu, ps, Hq,
                         max(ext), max(seq))
    assert torch.isfinite(o.float()).all()
    assert torch.allclose(o.float(), ref, rtol=BF16_RTOL, atol=BF16_ATOL), float((o.float() - ref).abs().max())


@pytest.mark.parametrize("Hq,Hkv,D,ps", [(32, 8, 128, 16), (32, 8, 64, 16), (8, 1, 128, 1), (16, 4, 128, 12)])
def test_prefill_short_turns_over_long_histories(Hq, Hkv, D, ps):
    """A few new tokens per request over thousands of cached ones (RadixCache hit of a long chat history): few row
    blocks, long KV - the launch is cut into KV splits whose fp32 partials a second kernel merges.  Includes a request
    with no prefix, one whose extend is a single token, and split counts that leave some splits empty."""
    m, ops = _mods()
    pre = [4096, 6000, 0, 2048, 515]
    ext = [40, 1, 70, 17, 130]
    pre = [p // ps * ps for p in pre]
    seq = [p + e for p, e in zip(pre, ext)]
    B = len(seq)
    table, rows, kp, vp = _random_paged_case(B, Hq, Hkv, D, ps, seq, seed=D + ps)
    T = sum(ext)
    assert ops.prefill_workspace_bytes(T, Hq, D, max(seq)) > 0          # this shape is split
    g = torch.Generator().manual_seed(2)
    q = torch.randn((T, Hq * D), generator=g).to(torch.bfloat16).to(DEV)
    dk, dv = kp.to(DEV), vp.to(DEV)
    dt, dr = torch.from_numpy(table).to(DEV), torch.from_numpy(rows).to(DEV)
    ref = _gpu_prefill_reference(q, dk, dv, dt, rows.tolist(), seq, pre, ps, Hq)
    cu = torch.tensor(np.concatenate([[0], np.cumsum(ext)]), dtype=torch.int32, device=DEV)
    o = ops.prefill_attn(q, dk, dv, dt, dr, torch.tensor(seq, device=DEV), torch.tensor(pre, device=DEV), cu, ps, Hq,
                         max(ext), max(seq))
    assert torch.isfinite(o.float()).all()
    assert torch.allclose(o.float(), ref, rtol=BF16_RTOL, atol=BF16_ATOL), float((o.float() - ref).abs().max())


def test_decode_cfg2_full_size_properties():
    """BASELINE.json configs[1] at full size (1B heads, B=64 ragged 146..2046, ps=16, shuffled pages):
    the result must not depend on the chunking, and must match a torch fp32 computation on the GPU."""
    import random
    m, ops = _mods()
    Hq, Hkv, D, ps = 32, 8, 64, 16
    random.seed(0)
    lens = [random.randint(64, 2048) for _ in range(64)]
    assert sum(lens) == 73961
    lens = [l + 1 for l in lens]
    B = len(lens)
    g = torch.Generator(device=DEV).manual_seed(0)
    npages = sum(-(-l // ps) for l in lens)
    perm = torch.randperm(npages, generator=torch.Generator().manual_seed(0))
    W = 8192 // ps
    table = torch.zeros((B, W), dtype=torch.int32)
    off = 0
    for b, l in enumerate(lens):
        n = -(-l // ps)
        table[b, :n] = perm[off:off + n].to(torch.int32)
        off += n
    # rows padded as mi355_attn.KVCache pads them (the headline's layout: the share kernel), and the reference's contiguous
    # rows (head_dim 64: the range kernel + merge)
    kp = _padded(torch.randn((npages * ps, Hkv, D), generator=g, device=DEV, dtype=torch.float32).to(torch.bfloat16))
    vp = _padded(torch.randn((npages * ps, Hkv, D), generator=g, device=DEV, dtype=torch.float32).to(torch.bfloat16))
    pitch = ops.pool_pitch(kp, vp)
    q = torch.randn((B, Hq * D), generator=g, device=DEV, dtype=torch.float32).to(torch.bfloat16)
    dt, dr, dl = table.to(DEV), torch.arange(B, device=DEV), torch.tensor(lens, device=DEV)
    outs = [ops.decode_attn(q, kp, vp, dt, dr, dl, ps, Hq, hint).float() for hint in (max(lens), 2 * max(lens))]
    plan = ops.decode_plan(dt, dr, dl, ps, Hq, Hkv, D, sum(lens), pool_row_stride=pitch)
    outs.append(ops.decode_attn_planned(q, kp, vp, plan, Hq).float())
    assert ops.decode_planned_workspace_bytes(B, Hq, Hkv, D, sum(lens), pitch) == 0 and plan.error() == 0     # the share kernel, no give-up
    kc, vc = kp.contiguous(), vp.contiguous()
    assert ops.decode_planned_workspace_bytes(B, Hq, Hkv, D, sum(lens), 0) > 0                            # contiguous: ranges + merge
    outs.append(ops.decode_attn_planned(q, kc, vc, ops.decode_plan(dt, dr, dl, ps, Hq, Hkv, D, sum(lens)), Hq).float())
    outs.append(ops.decode_attn(q, kc, vc, dt, dr, dl, ps, Hq, max(lens)).float())
    for o in outs[1:]:
        assert torch.allclose(o, outs[0], rtol=BF16_RTOL, atol=BF16_ATOL)
    # torch fp32 reference on the GPU, one sequence at a time (same gather as the reference backend)
    ref = torch.empty((B, Hq, D), device=DEV)
    for b, l in enumerate(lens):
        pos = torch.arange(l, device=DEV)
        slots = dt[b, pos // ps].long() * ps + pos % ps
        k = kp[slots].float().repeat_interleave(Hq // Hkv, dim=1)      # [L, Hq, D]
        v = vp[slots].float().repeat_interleave(Hq // Hkv, dim=1)
        s = torch.einsum("hd,lhd->hl", q[b].view(Hq, D).float(), k) * D ** -0.5
        ref[b] = torch.einsum("hl,lhd->hd", torch.softmax(s, dim=-1), v)
    assert torch.allclose(outs[0], ref.view(B, -1), rtol=BF16_RTOL, atol=BF16_ATOL)


def test_random_configurations_sweep():
    """tools/stress_parity.py for a few seconds: random head shapes, page sizes, batch compositions and prefix
    splits through planned decode with fused store, the self-contained decode call and prefill, against torch fp32 on the GPU."""
    import subprocess
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "stress_parity.py")
    # bf16 / IEEE half on the product library (it picks the decode kernel from the batch shape), then the dev build with
    # each decode kernel FORCED onto every batch (M355_DECODE_SHARE), so both see shapes the rule would not give them
    from mi355_attn import _lib
    runs = [([], {}), (["f16"], {})]
    if os.path.exists(_lib.DEV_LIB_PATH):
        runs += [(x, {"M355_LIB_PATH": "dev", "M355_DECODE_SHARE": s}) for s in ("0", "1") for x in ([], ["f16"])]
        # ... and the dev-only 64-rows-per-wave prefill kernel on every head_dim-128 extend (csrc/prefill_wg2.hip)
        runs += [([], {"M355_LIB_PATH": "dev", "M355_PREFILL_RT": "2"})]
        # ... and the dev-only in-kernel merge of the range kernel (decode_mfma.hip: last arriver merges; one and two levels)
        runs += [(x, {"M355_LIB_PATH": "dev", "M355_DECODE_SHARE": "0", "M355_MERGE_IN_KERNEL": "1"}) for x in ([], ["f16"])]
    for extra, env in runs:
        r = subprocess.run([sys.executable, tool, "1", "5"] + extra, capture_output=True, text=True, timeout=110,
                           env=dict(os.environ, **env))
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
        assert "random configurations passed" in r.stdout
