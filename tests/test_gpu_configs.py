"""GPU parity at BASELINE.json's own workloads and through the objects the reference hands a backend.

  cfg1  Llama-3.2-1B shape, page_size 1, batch 1, 128-token prompt + 16 decode steps (also the golden flow ps1_1b)
  cfg3  Llama-3-8B shape, 4 groups x 4 requests over a shared 1024-token cached prefix, at full size, then 4 decode steps
  cfg5  Llama-3-70B / TP8 per-rank shape (8, 1, 128), every request 32 768 tokens, B in {1, 8, 32}
  the reference's own Batch / KVCache shapes (no host mirrors, separate pool tensors per layer)
  PagedRadixCache -> prefix_ppns -> shared physical pages -> HIP prefill
  the device page ring at a batch larger than one round of its kernel

Bars as in test_gpu_parity.py: integers and pools bit-exact, attention within BF16_ATOL / BF16_RTOL of the oracle."""
import json
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from flow_util import BF16_ATOL, BF16_RTOL, GOLDEN, FlowReplay
from synth import synth
from oracle import attention_ref as A
from oracle import memory_ref as M

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _mods():
    import mi355_attn
    from mi355_attn import ops
    return mi355_attn, ops


def _runner(m, ps, kv, pm):
    return SimpleNamespace(device=DEV, page_size=ps, kvcache=kv, page_manager=pm)


# ------------------------------------------------------------------------------------------------
# cfg1
# ------------------------------------------------------------------------------------------------
def test_cfg1_1b_ps1_prompt128_then_16_decode_steps():
    """configs[0] on the HIP path: one 128-token prompt prefilled with no prefix at page_size 1 (every token its own
    page), then 16 decode steps, through Batch + MI355AttnBackend; oracle = the fp64 formula on the oracle's pools."""
    m, ops = _mods()
    Hq, Hkv, D, ps, L0, steps = 32, 8, 64, 1, 128, 16
    n_pages = L0 + steps + 8
    perm = torch.randperm(n_pages, generator=torch.Generator().manual_seed(3)).tolist()     # scattered pages
    kv = m.KVCache(size=n_pages, page_size=ps, head_num=Hkv, head_dim=D, dtype=torch.bfloat16, layer_num=1, device=DEV,
                   device_free_list=True)
    kv.set_free_pages(perm)
    pm = m.PageManager(page_size=ps, max_req_num=8, max_context_len=L0 + steps + 4, device=DEV)
    backend = m.MI355AttnBackend(_runner(m, ps, kv, pm))
    layer = m.Attention(layer_id=0, num_heads=Hq, num_kv_heads=Hkv, head_dim=D, scale=D ** -0.5)
    pool = M.RefPagePool(n_pages, ps)
    pool.free_pages = list(perm)
    pt = M.RefPageTable(ps, 8, L0 + steps + 4)
    rk = torch.zeros((n_pages, Hkv, D), dtype=torch.bfloat16)
    rv = torch.zeros_like(rk)

    req = m.Req(0, list(range(L0)))                       # input_ids = arange(128)
    req.fill_ids = list(req.origin_input_ids)
    batch = m.Batch([req], page_manager=pm, kvcache=kv, device=DEV)
    batch.prepare_for_prefill()
    meta = M.prepare_for_prefill_ref([M.RefReq(fill_ids=list(range(L0)))], pool, pt)
    for step in range(-1, steps):
        if step >= 0:
            batch.output_ids = torch.tensor([1000 + step], dtype=torch.int64, device=DEV)
            batch.prepare_for_decode()
            meta = M.prepare_for_decode_ref(meta, [1000 + step], pool, pt)
        for f in ("seq_lens", "positions", "out_cache_loc", "page_table_ids"):
            assert np.array_equal(getattr(batch, f).cpu().numpy(), getattr(meta, f)), (step, f)
        T = int(meta.input_ids.shape[0])
        qkv = synth((T, (Hq + 2 * Hkv) * D), 50 + step, amp=2.0, dtype=torch.bfloat16)
        q, k, v = qkv.split([Hq * D, Hkv * D, Hkv * D], dim=-1)
        A.write_kv_ref(rk, rv, torch.from_numpy(meta.out_cache_loc), k.view(-1, Hkv, D), v.view(-1, Hkv, D))
        ref = A.paged_attn_f64(q.reshape(T, Hq, D), rk, rv, torch.from_numpy(pt.table), torch.from_numpy(meta.page_table_ids),
                               torch.from_numpy(meta.seq_lens), torch.from_numpy(meta.prefix_lens) if step < 0 else None,
                               ps).reshape(T, Hq * D).float().numpy()
        dq, dk, dv = qkv.to(DEV).split([Hq * D, Hkv * D, Hkv * D], dim=-1)
        backend.init_forward_metadata(batch)
        o = backend.forward(dq, dk, dv, layer, batch)
        torch.cuda.synchronize()
        kp, vp = kv.get_kv_cache(0)
        assert torch.equal(kp.cpu(), rk) and torch.equal(vp.cpu(), rv), step
        np.testing.assert_allclose(o.float().cpu().numpy(), ref, rtol=BF16_RTOL, atol=BF16_ATOL, err_msg=f"step {step}")


# ------------------------------------------------------------------------------------------------
# group sizes the kernels are not built for (the reference takes any Llama config, utils/model_config.py:24-47)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("Hq,Hkv,D", [(24, 8, 128), (6, 2, 64), (7, 1, 128), (5, 1, 64)],
                         ids=["llama32_3b_G3", "G3_D64", "yi34b_tp8_G7", "G5"])
def test_group_sizes_outside_1_2_4_8(Hq, Hkv, D):
    """Llama-3.2-3B has 24 query heads on 8 kv heads (G = 3), Yi-34B 56 on 8 (G = 7): the backend runs such layers in the
    next built group size with zero query heads appended.  Three ragged requests prefilled at page_size 16, then 4 decode
    steps across a page boundary, through Batch + MI355AttnBackend; pools bit-exact, outputs against the fp64 oracle."""
    m, ops = _mods()
    ps, lens0, steps = 16, [70, 31, 130], 4
    n_pages = sum(-(-(l + steps) // ps) for l in lens0) + 8
    perm = torch.randperm(n_pages, generator=torch.Generator().manual_seed(5)).tolist()
    kv = m.KVCache(size=n_pages * ps, page_size=ps, head_num=Hkv, head_dim=D, dtype=torch.bfloat16, layer_num=1, device=DEV,
                   device_free_list=True)
    kv.set_free_pages(perm)
    ctx = max(lens0) + steps + ps
    pm = m.PageManager(page_size=ps, max_req_num=8, max_context_len=ctx, device=DEV)
    backend = m.MI355AttnBackend(_runner(m, ps, kv, pm))
    layer = m.Attention(layer_id=0, num_heads=Hq, num_kv_heads=Hkv, head_dim=D, scale=D ** -0.5)
    pool = M.RefPagePool(n_pages, ps)
    pool.free_pages = list(perm)
    pt = M.RefPageTable(ps, 8, ctx)
    rk = torch.zeros((n_pages * ps, Hkv, D), dtype=torch.bfloat16)
    rv = torch.zeros_like(rk)
    reqs = []
    for i, l in enumerate(lens0):
        r = m.Req(i, list(range(l)))
        r.fill_ids = list(r.origin_input_ids)
        reqs.append(r)
    batch = m.Batch(reqs, page_manager=pm, kvcache=kv, device=DEV)
    batch.prepare_for_prefill()
    meta = M.prepare_for_prefill_ref([M.RefReq(fill_ids=list(range(l))) for l in lens0], pool, pt)
    for step in range(-1, steps):
        if step >= 0:
            nxt = [1000 + step + i for i in range(len(lens0))]
            batch.output_ids = torch.tensor(nxt, dtype=torch.int64, device=DEV)
            batch.prepare_for_decode()
            meta = M.prepare_for_decode_ref(meta, nxt, pool, pt)
        for f in ("seq_lens", "positions", "out_cache_loc", "page_table_ids"):
            assert np.array_equal(getattr(batch, f).cpu().numpy(), getattr(meta, f)), (step, f)
        T = int(meta.input_ids.shape[0])
        qkv = synth((T, (Hq + 2 * Hkv) * D), 70 + step, amp=2.0, dtype=torch.bfloat16)
        q, k, v = qkv.split([Hq * D, Hkv * D, Hkv * D], dim=-1)
        A.write_kv_ref(rk, rv, torch.from_numpy(meta.out_cache_loc), k.view(-1, Hkv, D), v.view(-1, Hkv, D))
        ref = A.paged_attn_f64(q.reshape(T, Hq, D), rk, rv, torch.from_numpy(pt.table), torch.from_numpy(meta.page_table_ids),
                               torch.from_numpy(meta.seq_lens), torch.from_numpy(meta.prefix_lens) if step < 0 else None,
                               ps).reshape(T, Hq * D).float().numpy()
        dq, dk, dv = qkv.to(DEV).split([Hq * D, Hkv * D, Hkv * D], dim=-1)      # strided views of the packed projection
        backend.init_forward_metadata(batch)
        o = backend.forward(dq, dk, dv, layer, batch)
        torch.cuda.synchronize()
        assert o.shape == (T, Hq * D) and o.dtype == torch.bfloat16
        kp, vp = kv.get_kv_cache(0)
        assert torch.equal(kp.cpu(), rk) and torch.equal(vp.cpu(), rv), step
        np.testing.assert_allclose(o.float().cpu().numpy(), ref, rtol=BF16_RTOL, atol=BF16_ATOL, err_msg=f"step {step}")


# ------------------------------------------------------------------------------------------------
# cfg5
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B", [1, 8])
def test_cfg5_70b_tp8_32k_decode(B):
    """configs[4] per rank: (Hq, Hkv, D) = (8, 1, 128), page_size 16, every request 32 768 tokens (the new one
    included), planned decode with the KV store fused in; requests are cut into many pieces here (the wide merge at
    B = 1).  Oracle: paged_attn_f64 on the CPU copy of the pools after the write."""
    m, ops = _mods()
    Hq, Hkv, D, ps, L = 8, 1, 128, 16, 32768
    lens = [L] * B
    npages = B * (L // ps)
    perm = torch.randperm(npages, generator=torch.Generator().manual_seed(B)).to(torch.int32)
    table = perm.view(B, L // ps).contiguous()
    g = torch.Generator().manual_seed(100 + B)
    kp = torch.randn((npages * ps, Hkv, D), generator=g).to(torch.bfloat16)
    vp = torch.randn((npages * ps, Hkv, D), generator=g).to(torch.bfloat16)
    qkv = (torch.randn((B, (Hq + 2 * Hkv) * D), generator=g) * 2).to(torch.bfloat16)
    q, k, v = qkv.split([Hq * D, Hkv * D, Hkv * D], dim=-1)
    rows = torch.arange(B)
    pos = np.asarray(lens, dtype=np.int64) - 1
    loc = torch.from_numpy(M.out_cache_loc_ref(table.numpy(), rows.numpy(), pos, ps))
    rk, rv = kp.clone(), vp.clone()
    A.write_kv_ref(rk, rv, loc, k.reshape(B, Hkv, D), v.reshape(B, Hkv, D))
    ref = A.paged_attn_f64(q.reshape(B, Hq, D), rk, rv, table, rows, torch.tensor(lens), None, ps).reshape(B, Hq * D).float().numpy()

    dqkv = qkv.to(DEV)
    dq, dk, dv = dqkv.split([Hq * D, Hkv * D, Hkv * D], dim=-1)
    dkp, dvp, dt = kp.to(DEV), vp.to(DEV), table.to(DEV)
    dr, dl = rows.to(DEV), torch.tensor(lens, device=DEV)
    plan = ops.decode_plan(dt, dr, dl, ps, Hq, Hkv, D, sum(lens))
    o = ops.decode_attn_planned(dq, dkp, dvp, plan, Hq, k_new=dk, v_new=dv)
    torch.cuda.synchronize()
    assert torch.equal(dkp.cpu(), rk) and torch.equal(dvp.cpu(), rv)
    np.testing.assert_allclose(o.float().cpu().numpy(), ref, rtol=BF16_RTOL, atol=BF16_ATOL)
    # the plan-free form of the same pools gives the same answer
    o2 = ops.decode_attn_planned(dq, dkp, dvp, plan, Hq)
    torch.cuda.synchronize()
    np.testing.assert_allclose(o2.float().cpu().numpy(), ref, rtol=BF16_RTOL, atol=BF16_ATOL)
    assert plan.error() == 0


def test_cfg5_70b_tp8_32k_decode_B32_share_kernel():
    """configs[4] per rank at B = 32: the size bench.py quotes for cfg5 (1 M cached tokens, 512 MiB of K/V per layer).  The
    library serves it with the single-launch SHARE kernel, every request spanning 8 shares - 7 publishers hand their
    partials to each merging workgroup - which no smaller test reaches.  Reference: torch fp32 on the GPU, one request
    at a time, same gather as the reference backend (torch_native_backend.py:171-186); pools checked against
    store-then-attend bit for bit."""
    m, ops = _mods()
    from test_gpu_parity import _gpu_decode_reference
    Hq, Hkv, D, ps, L, B = 8, 1, 128, 16, 32768, 32
    lens = [L] * B
    assert ops.decode_planned_workspace_bytes(B, Hq, Hkv, D, sum(lens)) == 0, "the share kernel is the one measured at B = 32"
    npages = B * (L // ps)
    table = torch.randperm(npages, generator=torch.Generator().manual_seed(B)).to(torch.int32).view(B, L // ps).contiguous()
    g = torch.Generator(device=DEV).manual_seed(100 + B)
    kp = torch.randn((npages * ps, Hkv, D), generator=g, device=DEV).to(torch.bfloat16)
    vp = torch.randn((npages * ps, Hkv, D), generator=g, device=DEV).to(torch.bfloat16)
    qkv = (torch.randn((B, (Hq + 2 * Hkv) * D), generator=g, device=DEV) * 2).to(torch.bfloat16)
    q, k, v = qkv.split([Hq * D, Hkv * D, Hkv * D], dim=-1)
    dt, dr, dl = table.to(DEV), torch.arange(B, device=DEV), torch.tensor(lens, device=DEV)
    loc = ops.out_cache_loc(dt, dr, dl - 1, ps)
    kp1, vp1 = kp.clone(), vp.clone()
    ops.kv_store(kp1, vp1, loc, k, v)
    ref = _gpu_decode_reference(q.contiguous(), kp1, vp1, dt, list(range(B)), lens, ps, Hq)
    plan = ops.decode_plan(dt, dr, dl, ps, Hq, Hkv, D, sum(lens))
    for rep in range(3):                                  # the hand-off granules are lowered and raised again every launch
        o = ops.decode_attn_planned(q, kp, vp, plan, Hq, k_new=k, v_new=v)
        torch.cuda.synchronize()
        assert torch.equal(kp, kp1) and torch.equal(vp, vp1), rep
        assert torch.allclose(o.float(), ref, rtol=BF16_RTOL, atol=BF16_ATOL), (rep, float((o.float() - ref).abs().max()))
    assert plan.error() == 0


# ------------------------------------------------------------------------------------------------
# cfg3
# ------------------------------------------------------------------------------------------------
def test_cfg3_full_size_prefill_then_decode():
    """configs[2] at the size bench.py times (harness.workloads.prefill_cfg3_case): Llama-3-8B shape (32, 8, 128), 16
    requests in 4 groups, each group's 1024-token prefix on the SAME physical pages (what RadixCache hits give), 128..512
    new tokens each - 1216 row blocks in one launch - and then 4 decode steps of the same batch.  Prefill against torch
    fp32 on the GPU (row j sees cached tokens 0 .. prefix + j, torch_native_backend.py:25-119); every decode step against
    the oracle's gather-and-softmax restated in torch fp32 on the GPU, pools against store-then-attend bit for bit."""
    m, ops = _mods()
    from harness import workloads as W
    from test_gpu_parity import _gpu_decode_reference, _gpu_prefill_reference
    c = W.prefill_cfg3_case(DEV, layers=1)
    kp, vp = c.pools[0]
    o = ops.prefill_attn(c.q, kp, vp, c.table, c.rows, c.dseq, c.dpre, c.cu, c.ps, c.Hq, max(c.ext), max(c.seq))
    torch.cuda.synchronize()
    ref = _gpu_prefill_reference(c.q, kp, vp, c.table, list(range(c.B)), c.seq, c.pre, c.ps, c.Hq)
    assert torch.isfinite(o.float()).all()
    assert torch.allclose(o.float(), ref, rtol=BF16_RTOL, atol=BF16_ATOL), float((o.float() - ref).abs().max())
    # ---- 4 decode steps: every request grows by one token per step (its last page has room, or takes a fresh page)
    Hq, Hkv, D, ps, B = c.Hq, c.Hkv, c.D, c.ps, c.B
    table = torch.nn.functional.pad(c.table, (0, 2))
    next_page = int(kp.shape[0] // ps)                     # fresh pages: B more rows of pages appended to the pools
    g = torch.Generator(device=DEV).manual_seed(4)
    kp = torch.cat([kp, torch.randn((B * ps, Hkv, D), generator=g, device=DEV).to(torch.bfloat16)])
    vp = torch.cat([vp, torch.randn((B * ps, Hkv, D), generator=g, device=DEV).to(torch.bfloat16)])
    lens = list(c.seq)
    g = torch.Generator(device=DEV).manual_seed(5)
    rows = torch.arange(B, device=DEV)
    for step in range(4):
        for b in range(B):
            if lens[b] % ps == 0:
                assert next_page < kp.shape[0] // ps
                table[b, lens[b] // ps] = next_page
                next_page += 1
            lens[b] += 1
        dl = torch.tensor(lens, device=DEV)
        qkv = (torch.randn((B, (Hq + 2 * Hkv) * D), generator=g, device=DEV) * 2).to(torch.bfloat16)
        q, k, v = qkv.split([Hq * D, Hkv * D, Hkv * D], dim=-1)
        loc = ops.out_cache_loc(table, rows, dl - 1, ps)
        kp1, vp1 = kp.clone(), vp.clone()
        ops.kv_store(kp1, vp1, loc, k, v)
        refd = _gpu_decode_reference(q.contiguous(), kp1, vp1, table, list(range(B)), lens, ps, Hq)
        plan = ops.decode_plan(table, rows, dl, ps, Hq, Hkv, D, sum(lens))
        od = ops.decode_attn_planned(q, kp, vp, plan, Hq, k_new=k, v_new=v)
        torch.cuda.synchronize()
        assert torch.equal(kp, kp1) and torch.equal(vp, vp1), step
        assert torch.allclose(od.float(), refd, rtol=BF16_RTOL, atol=BF16_ATOL), (step, float((od.float() - refd).abs().max()))
        assert plan.error() == 0


# ------------------------------------------------------------------------------------------------
# the reference's own objects
# ------------------------------------------------------------------------------------------------
class _RefShapedKVCache:
    """memory/kvcache.py:24-35, 81-85: one separate tensor per layer for K and for V, nothing else the backend may use
    except head_num / head_dim / page_size."""

    def __init__(self, rows, Hkv, D, layers, ps):
        self.page_size, self.head_num, self.head_dim, self.layer_num = ps, Hkv, D, layers
        self.page_num = rows // ps
        self.k_cache = [torch.zeros((rows, Hkv, D), dtype=torch.bfloat16, device=DEV) for _ in range(layers)]
        self.v_cache = [torch.zeros((rows, Hkv, D), dtype=torch.bfloat16, device=DEV) for _ in range(layers)]

    def get_kv_cache(self, layer_id):
        return self.k_cache[layer_id], self.v_cache[layer_id]


class _RefMode:
    def __init__(self, decode):
        self._d = decode

    def is_decode(self):
        return self._d

    def is_prefill(self):
        return not self._d


@pytest.mark.parametrize("name", ["ps16_1b", "ps1_8b_tp2", "ps4_tiny", "ps16_70b_tp8"])
def test_backend_with_reference_shaped_batch(name):
    """What INTEGRATION.md promises: the backend driven by objects that carry only what the reference's Batch,
    KVCache and PageManager carry (engine/batch.py:192-231, memory/kvcache.py:24-35): device tensors mode / seq_lens /
    prefix_lens / page_table_ids / out_cache_loc, no seq_lens_host, no cu_q, k and v passed, one pool tensor per layer,
    keyword arguments as layers/attention.py:30-36 passes them.  A golden flow is replayed through it, with the
    reference's own metadata values; pools bit-exact, outputs within tolerance, on a layer other than 0."""
    m, ops = _mods()
    fr = FlowReplay(os.path.join(GOLDEN, f"flow_{name}.npz"), torch.bfloat16)
    ps, Hq, Hkv, D = fr.ps, fr.Hq, fr.Hkv, fr.D
    kv = _RefShapedKVCache(fr.S_pages * ps, Hkv, D, layers=2, ps=ps)
    pm = SimpleNamespace(page_size=ps, page_table=torch.zeros((fr.max_req, fr.ctx // ps), dtype=torch.int32, device=DEV))
    backend = m.MI355AttnBackend(SimpleNamespace(device=DEV, page_size=ps))            # torch_native_backend.py:16-19 reads only these
    layer = SimpleNamespace(layer_id=1, num_heads=Hq, num_kv_heads=Hkv, head_dim=D, scale=None)
    for ph in fr.phases():
        pm.page_table.copy_(torch.from_numpy(ph.table))               # as Batch.prepare_* left it
        batch = SimpleNamespace(
            mode=_RefMode(ph.decode), reqs=[None] * fr.n, page_manager=pm, page_size=ps, kvcache=kv, device=DEV,
            seq_lens=torch.from_numpy(fr.golden(ph.tag, "seq_lens")).to(DEV),
            prefix_lens=torch.from_numpy(fr.golden(ph.tag, "prefix_lens")).to(DEV),
            page_table_ids=torch.from_numpy(fr.golden(ph.tag, "page_table_ids")).to(DEV),
            out_cache_loc=torch.from_numpy(fr.golden(ph.tag, "out_cache_loc")).to(DEV),
            positions=torch.from_numpy(fr.golden(ph.tag, "positions")).to(DEV))
        qkv = ph.qkv.to(DEV)
        q, k, v = qkv.split([Hq * D, Hkv * D, Hkv * D], dim=-1)
        backend.init_forward_metadata(batch)                          # model_runner.py:194
        batch.attn_backend = backend                                  # :195
        o = batch.attn_backend.forward(q=q, k=k, v=v, layer=layer, batch=batch)
        torch.cuda.synchronize()
        assert o.shape == (q.shape[0], Hq * D) and o.dtype == torch.bfloat16 and o.data_ptr() != q.data_ptr()
        assert torch.equal(kv.k_cache[1].cpu(), ph.k_pool) and torch.equal(kv.v_cache[1].cpu(), ph.v_pool), ph.tag
        assert not kv.k_cache[0].any() and not kv.v_cache[0].any()   # the other layer's pools are untouched
        rows_, ref = fr.golden_o_f32(ph.tag)
        np.testing.assert_allclose(o.float().cpu().numpy()[rows_], ref, rtol=BF16_RTOL, atol=BF16_ATOL, err_msg=ph.tag)


def test_decode_steps_without_the_metadata_hook():
    """two consecutive decode forwards with NO init_forward_metadata in between (legal against the
    torch-native backend, whose hook is a no-op).  The per-forward plan must follow the batch: the second step attends
    the second step's token and writes the second step's slots."""
    m, ops = _mods()
    fr = FlowReplay(os.path.join(GOLDEN, "flow_ps4_tiny.npz"), torch.bfloat16)
    ps, Hq, Hkv, D = fr.ps, fr.Hq, fr.Hkv, fr.D
    kv = m.KVCache(size=fr.S_pages * ps, page_size=ps, head_num=Hkv, head_dim=D, dtype=torch.bfloat16, layer_num=2, device=DEV)
    pm = m.PageManager(page_size=ps, max_req_num=fr.max_req, max_context_len=fr.ctx, device=DEV)
    backend = m.MI355AttnBackend(_runner(m, ps, kv, pm))
    layers = [m.Attention(layer_id=i, num_heads=Hq, num_kv_heads=Hkv, head_dim=D, scale=D ** -0.5) for i in range(2)]
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
        qkv = ph.qkv.to(DEV)
        q, k, v = qkv.split([Hq * D, Hkv * D, Hkv * D], dim=-1)
        if not ph.decode or ph.tag == "d0":
            backend.init_forward_metadata(batch)          # prefills and the first decode step: hook called; later: not
        batch.attn_backend = backend
        outs = [layer(q, k, v, batch) for layer in layers]            # two layers per forward
        torch.cuda.synchronize()
        for li, o in enumerate(outs):
            kp, vp = kv.get_kv_cache(li)
            assert torch.equal(kp.cpu(), ph.k_pool) and torch.equal(vp.cpu(), ph.v_pool), (ph.tag, li)
            np.testing.assert_allclose(o.float().cpu().numpy(), ph.out.float().numpy(), rtol=BF16_RTOL, atol=BF16_ATOL,
                                       err_msg=f"{ph.tag} layer {li}")


# ------------------------------------------------------------------------------------------------
# radix cache -> shared physical pages -> prefill
# ------------------------------------------------------------------------------------------------
def _token_kv(ids, pos, Hq, Hkv, D, salt):
    """Deterministic q/k/v rows for (token id, position): what a model would produce - equal tokens at equal positions
    give equal K/V, so a shared cached page holds exactly what a request that re-computed it would have written."""
    rows = []
    for t, p in zip(ids, pos):
        rows.append(synth((1, (Hq + 2 * Hkv) * D), 7919 * int(t) + 31 * int(p) + salt, amp=2.0, dtype=torch.bfloat16))
    return torch.cat(rows, 0)


@pytest.mark.parametrize("native", [False, True], ids=["python_tree", "native_tree"])
def test_radix_hits_feed_shared_pages_to_prefill(native):
    """north_star: "ragged prefill over RadixCache-hit prefixes".  The scheduler's sequence (engine/scheduler.py:295-297,
    415-436; engine/batch.py:128-132) on the golden trace ps16_shared_prefix: match_prefix -> prefix_ppns ->
    prepare_for_prefill -> HIP KV store + prefill over pages that EARLIER requests wrote -> cache_unfinished_req.
    Page rows, prefixes and slots must be the reference's (radix_traces.json); attention is checked against the oracle
    on CPU pools that only ever receive each request's NEW tokens.  Then two fresh requests that share the cached
    prefix go through ONE batch (the same physical pages in two page-table rows of one launch)."""
    m, ops = _mods()
    from mi355_attn.radix_cache import NativeRadixCache, PagedRadixCache
    with open(os.path.join(GOLDEN, "radix_traces.json")) as f:
        tr = json.load(f)["ps16_shared_prefix"]
    ps, prompts = tr["ps"], tr["prompts"]
    Hq, Hkv, D = 8, 2, 128
    ctx = 256
    kv = m.KVCache(size=tr["n_pages"] * ps, page_size=ps, head_num=Hkv, head_dim=D, dtype=torch.bfloat16, layer_num=1, device=DEV)
    pm = m.PageManager(page_size=ps, max_req_num=len(prompts) + 3, max_context_len=ctx, device=DEV)
    tree = (NativeRadixCache if native else PagedRadixCache)(pm, kv, reference_split_walk=True)
    backend = m.MI355AttnBackend(_runner(m, ps, kv, pm))
    layer = m.Attention(layer_id=0, num_heads=Hq, num_kv_heads=Hkv, head_dim=D, scale=D ** -0.5)
    rk = torch.zeros((tr["n_pages"] * ps, Hkv, D), dtype=torch.bfloat16)
    rv = torch.zeros_like(rk)
    golden = [s for s in tr["steps"] if s["op"] == "prefill"]

    def serve(reqs):
        for r in reqs:
            r.init_next_round_input(tree)                              # match_prefix (batch.py:128-132)
        batch = m.Batch(reqs, page_manager=pm, kvcache=kv, tree_cache=tree, device=DEV)
        batch.prepare_for_prefill()
        pre = [len(r.prefix_ppns) * ps for r in reqs]
        ids = [r.fill_ids[p:] for r, p in zip(reqs, pre)]
        qkv = torch.cat([_token_kv(i_, range(p, p + len(i_)), Hq, Hkv, D, 5) for i_, p in zip(ids, pre)], 0)
        loc = batch.out_cache_loc.cpu()
        q, k, v = qkv.split([Hq * D, Hkv * D, Hkv * D], dim=-1)
        A.write_kv_ref(rk, rv, loc, k.reshape(-1, Hkv, D), v.reshape(-1, Hkv, D))          # only the NEW tokens
        table = pm.page_table.cpu()
        ref = A.paged_attn_f64(q.reshape(-1, Hq, D), rk, rv, table, batch.page_table_ids.cpu(), batch.seq_lens.cpu(),
                               batch.prefix_lens.cpu(), ps).reshape(-1, Hq * D).float().numpy()
        dq, dk, dv = qkv.to(DEV).split([Hq * D, Hkv * D, Hkv * D], dim=-1)
        backend.init_forward_metadata(batch)
        o = backend.forward(dq, dk, dv, layer, batch)
        torch.cuda.synchronize()
        kp, vp = kv.get_kv_cache(0)
        assert torch.equal(kp.cpu(), rk) and torch.equal(vp.cpu(), rv)
        np.testing.assert_allclose(o.float().cpu().numpy(), ref, rtol=BF16_RTOL, atol=BF16_ATOL)
        return batch, pre

    reqs = []
    for i, p in enumerate(prompts):
        r = m.Req(i, list(p), m.SamplingParams(max_new_tokens=4))
        batch, pre = serve([r])
        g = golden[i]
        used = -(-len(p) // ps)
        assert [int(x) for x in r.prefix_ppns] == g["prefix_ppns"] and pre[0] == g["prefix_len"]
        assert pm.page_table[r.page_table_id, :used].tolist() == g["page_row"]
        assert batch.out_cache_loc.tolist() == g["out_cache_loc"]
        r.output_ids.append(7000 + i)
        tree.cache_unfinished_req(r)
        reqs.append(r)
    assert golden[1]["prefix_len"] > 0 and golden[2]["prefix_len"] > 0         # the trace does hit the cache
    # two new requests over the cached 48-token system prompt, in one batch: both rows start with the SAME pages
    a = m.Req(10, list(range(48)) + [900 + j for j in range(37)], m.SamplingParams(max_new_tokens=4))
    b = m.Req(11, list(range(48)) + [950 + j for j in range(5)], m.SamplingParams(max_new_tokens=4))
    batch, pre = serve([a, b])
    assert pre == [48, 48]
    ra, rb = pm.page_table[a.page_table_id, :3].tolist(), pm.page_table[b.page_table_id, :3].tolist()
    assert ra == rb == [int(x) for x in reqs[0].prefix_ppns[:3]]


# ------------------------------------------------------------------------------------------------
# metadata kernels at sizes and edges the flows do not reach
# ------------------------------------------------------------------------------------------------
def test_decode_prepare_ring_more_than_one_round():
    """B = 1500 > the 1024 requests one round of decode_prepare_ring_kernel handles: the pages taken in round 1 must
    carry over into round 2 (engine/batch.py:325-343 + memory/kvcache.py:49-56 in batch order)."""
    m, ops = _mods()
    rng = np.random.default_rng(8)
    for ps in (1, 4, 16):
        B = 1500
        W = 12
        lens = rng.integers(1, (W - 1) * ps, size=B).astype(np.int64)
        lens[rng.integers(0, B, size=400)] = rng.integers(1, W - 1, size=400) * ps          # many page starts
        n_need = int((lens % ps == 0).sum())
        cap = 4096
        ring_pages = rng.permutation(cap)[: n_need + 50].astype(np.int32)
        table = rng.integers(0, 1 << 20, size=(B + 5, W), dtype=np.int32)
        rows = rng.permutation(B + 5)[:B].astype(np.int64)
        # host statement: the vectorised plan + the oracle's slot formula
        want_table = table.copy()
        new_pages = np.full(B, -1, dtype=np.int32)
        new_pages[lens % ps == 0] = ring_pages[:n_need]
        pos = lens.copy()
        for b in np.nonzero(new_pages >= 0)[0]:
            want_table[rows[b], pos[b] // ps] = new_pages[b]
        want_loc = M.out_cache_loc_ref(want_table, rows, pos, ps)
        ring = ops.PageRing(cap, DEV)
        ring.load(ring_pages.tolist())
        dt = torch.from_numpy(table).to(DEV)
        new_lens, positions, loc = ops.decode_prepare_ring(dt, torch.from_numpy(rows).to(DEV), torch.from_numpy(lens).to(DEV), ring, ps)
        torch.cuda.synchronize()
        assert np.array_equal(new_lens.cpu().numpy(), lens + 1)
        assert np.array_equal(positions.cpu().numpy(), pos)
        assert np.array_equal(loc.cpu().numpy(), want_loc), ps
        assert np.array_equal(dt.cpu().numpy(), want_table), ps
        assert ring.snapshot() == ring_pages[n_need:].tolist()
        # and the host-planned kernel agrees
        dt2 = torch.from_numpy(table).to(DEV)
        l2, p2, loc2 = ops.decode_prepare(dt2, torch.from_numpy(rows).to(DEV), torch.from_numpy(lens).to(DEV),
                                          torch.from_numpy(new_pages).to(DEV), ps)
        assert torch.equal(loc2, loc) and torch.equal(dt2, dt) and torch.equal(l2, new_lens)


def test_page_table_width_is_a_hard_bound():
    """a request longer than the page table is wide must not reach the next request's row.  The planner
    refuses it (nothing allocated); the kernels, called directly with such a plan, drop the out-of-range entries and
    mark the slots M355_NO_SLOT, which the store kernels skip (-1 wrapped to the pool's LAST slot); the decode
    plan cuts an over-long request at its row's end and leaves it out of the fused KV store."""
    m, ops = _mods()
    ps, W = 4, 3
    kv = m.KVCache(size=64 * ps, page_size=ps, head_num=1, head_dim=64, dtype=torch.bfloat16, layer_num=1, device=DEV)
    pm = m.PageManager(page_size=ps, max_req_num=4, max_context_len=W * ps, device=DEV)
    r = m.Req(0, list(range(W * ps + 1)))
    r.fill_ids = list(r.origin_input_ids)
    with pytest.raises(RuntimeError, match="columns"):
        m.Batch([r], page_manager=pm, kvcache=kv, device=DEV).prepare_for_prefill()
    assert kv.available_page_num() == 64 and len(pm.free_slots) == 4
    ok = m.Req(1, list(range(W * ps)))
    ok.fill_ids = list(ok.origin_input_ids)
    batch = m.Batch([ok], page_manager=pm, kvcache=kv, device=DEV)
    batch.prepare_for_prefill()                                       # exactly W pages: fine
    batch.output_ids = torch.zeros(1, dtype=torch.int64, device=DEV)
    with pytest.raises(RuntimeError, match="columns"):
        batch.prepare_for_decode()
    # the kernels themselves: 2 requests of 4 pages each into a table 3 columns wide
    table = torch.full((2, W), -7, dtype=torch.int32, device=DEV)
    plan = torch.tensor([[0, 0, 16, 0], [1, 0, 16, 4], [0, 0, 0, 8]], dtype=torch.int32, device=DEV).reshape(-1)
    pages = torch.arange(10, 18, dtype=torch.int32, device=DEV)
    cu_q = torch.tensor([0, 16, 32], dtype=torch.int32, device=DEV)
    pos, rot, loc = ops.prefill_prepare(table, plan, pages, cu_q, 2, 32, ps)
    torch.cuda.synchronize()
    assert table.tolist() == [[10, 11, 12], [14, 15, 16]]
    loc = loc.cpu().numpy().reshape(2, 16)
    NO_SLOT = ops._lib.NO_SLOT
    assert (loc[:, 12:] == NO_SLOT).all() and loc[0, :12].tolist() == list(range(40, 52)) and loc[1, :12].tolist() == list(range(56, 68))
    lens = torch.tensor([12, 5], dtype=torch.int64, device=DEV)
    l2, p2, loc2 = ops.decode_prepare(table, torch.tensor([0, 1], device=DEV), lens, torch.tensor([99, -1], dtype=torch.int32, device=DEV), ps)
    torch.cuda.synchronize()
    assert table.tolist() == [[10, 11, 12], [14, 15, 16]] and loc2.tolist() == [NO_SLOT, 15 * ps + 1]
    # prepare -> kv_store with the over-wide request: every pool row keeps its contents (the marked tokens are skipped;
    # -1 would have landed in the LAST slot, a live request's KV); same through the fused rope + store
    S, Hkv, D = 80, 1, 64
    kp0 = torch.randn((S, Hkv, D), device=DEV).to(torch.bfloat16)
    vp0 = torch.randn((S, Hkv, D), device=DEV).to(torch.bfloat16)
    kp, vp = kp0.clone(), vp0.clone()
    src = torch.randn((2, Hkv * D), device=DEV).to(torch.bfloat16)
    ops.kv_store(kp, vp, loc2, src, src)
    torch.cuda.synchronize()
    keep = torch.ones(S, dtype=torch.bool, device=DEV)
    keep[15 * ps + 1] = False
    assert torch.equal(kp[keep], kp0[keep]) and torch.equal(vp[keep], vp0[keep])
    assert torch.equal(kp[15 * ps + 1], src[1].view(Hkv, D))
    kp, vp = kp0.clone(), vp0.clone()
    cache = torch.ones((64, D), dtype=torch.float32, device=DEV)
    ops.rope_kv_store(torch.tensor([3, 4], device=DEV), src.clone(), src.clone(), D, cache, v=src, k_pool=kp, v_pool=vp, loc=loc2)
    torch.cuda.synchronize()
    assert torch.equal(kp[keep], kp0[keep]) and torch.equal(vp[keep], vp0[keep])
    # the decode plan: request 0 claims 13 tokens in a table 3 pages (12 tokens) wide.  It is attended over its 12
    # in-table tokens, nothing is stored for it, and request 1 (row 1, the row "behind" it) is served as usual.
    for Hq_ in (1, 4):
        kp, vp = kp0.clone(), vp0.clone()
        dl = torch.tensor([13, 6], dtype=torch.int64, device=DEV)
        rows_ = torch.tensor([0, 1], device=DEV)
        qkv = torch.randn((2, (Hq_ + 2) * D), device=DEV).to(torch.bfloat16)
        q, k, v = qkv.split([Hq_ * D, D, D], dim=-1)
        plan = ops.decode_plan(table, rows_, dl, ps, Hq_, Hkv, D, 19)
        o = ops.decode_attn_planned(q, kp, vp, plan, Hq_, k_new=k, v_new=v)
        torch.cuda.synchronize()
        slot1 = int(table[1, 1]) * ps + 1
        keep = torch.ones(S, dtype=torch.bool, device=DEV)
        keep[slot1] = False
        assert torch.equal(kp[keep], kp0[keep]) and torch.equal(vp[keep], vp0[keep])          # request 0 stored nothing
        assert torch.equal(kp[slot1], k[1].view(Hkv, D)) and torch.equal(vp[slot1], v[1].view(Hkv, D))
        from test_gpu_parity import _gpu_decode_reference
        ref = _gpu_decode_reference(q.contiguous(), kp, vp, table, [0, 1], [12, 6], ps, Hq_)
        assert torch.allclose(o.float(), ref, rtol=BF16_RTOL, atol=BF16_ATOL)
