"""CPU: the oracle against the golden vectors produced by the reference itself
(tests/golden/gen_golden.py).  This is what pins the oracle; the GPU tests then compare the HIP
path with the oracle and with the same fixtures."""
import json
import os
import warnings

import numpy as np
import pytest
import torch

from flow_util import BF16_ATOL, BF16_RTOL, FlowReplay, fixture_id, flow_fixtures
from synth import bits16, fnv64
from oracle import attention_ref as A
from oracle import memory_ref as M
from oracle import radix_ref as R

META_FIELDS = ("input_ids", "page_table_ids", "seq_lens", "prefix_lens", "positions", "out_cache_loc")


@pytest.fixture(autouse=True)
def _one_thread():
    n = torch.get_num_threads()
    torch.set_num_threads(1)
    yield
    torch.set_num_threads(n)


def test_fixtures_present():
    assert len(flow_fixtures()) >= 9


@pytest.mark.parametrize("path", flow_fixtures(), ids=fixture_id)
def test_flow_index_math_and_kv_write_bit_exact(path):
    fr = FlowReplay(path, torch.bfloat16)
    for ph in fr.phases():
        for f in META_FIELDS:
            got = getattr(ph.meta, f)
            assert got.dtype == np.int64
            assert np.array_equal(got, fr.golden(ph.tag, f)), (ph.tag, f)
        used = -(-int(ph.meta.seq_lens.max()) // fr.ps)
        assert np.array_equal(ph.table[ph.meta.page_table_ids, :used], fr.golden(ph.tag, "page_rows")), ph.tag
        assert ph.free_pages[:4] == [int(x) for x in fr.golden(ph.tag, "free_head")]
        assert fnv64(bits16(ph.k_pool)) == int(fr.golden(ph.tag, "k_pool_hash")), (ph.tag, "k pool")
        assert fnv64(bits16(ph.v_pool)) == int(fr.golden(ph.tag, "v_pool_hash")), (ph.tag, "v pool")


@pytest.mark.parametrize("path", flow_fixtures(), ids=fixture_id)
def test_flow_attention_fp32_matches_reference(path):
    fr = FlowReplay(path, torch.float32)
    for ph in fr.phases():
        rows, ref = fr.golden_o_f32(ph.tag)
        got = ph.out.numpy()[rows]
        # same torch ops in the same order; a different host ISA may reassociate a sum
        np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-6, err_msg=ph.tag)


@pytest.mark.parametrize("path", flow_fixtures(), ids=fixture_id)
def test_flow_attention_bf16_and_f64_anchor(path):
    fr = FlowReplay(path, torch.bfloat16)
    for ph in fr.phases():
        if fnv64(bits16(ph.out)) != int(fr.golden(ph.tag, "o_bf16_hash")):
            warnings.warn(f"{fixture_id(path)}:{ph.tag}: bf16 SDPA differs bitwise from the generating host")
        rows, ref = fr.golden_o_f32(ph.tag)
        np.testing.assert_allclose(ph.out.float().numpy()[rows], ref, rtol=BF16_RTOL, atol=BF16_ATOL, err_msg=ph.tag)
        # independent float64 textbook formula agrees with the fp32 reference output
        T = ph.q.shape[0]
        f64 = A.paged_attn_f64(ph.q.reshape(T, fr.Hq, fr.D), ph.k_pool, ph.v_pool, torch.from_numpy(ph.table),
                               torch.from_numpy(ph.meta.page_table_ids), torch.from_numpy(ph.meta.seq_lens),
                               None if ph.decode else torch.from_numpy(ph.meta.prefix_lens), fr.ps)
        np.testing.assert_allclose(f64.reshape(T, -1).numpy()[rows], ref, rtol=1e-4, atol=1e-5, err_msg=ph.tag)


def _f16_fixtures():
    return [p for p in flow_fixtures() if "p1.f16_o_hash" in np.load(p).files]


@pytest.mark.parametrize("path", _f16_fixtures(), ids=fixture_id)
def test_flow_in_ieee_half_matches_reference(path):
    """The flows generated a third time with IEEE-half pools and inputs (Llama-2-style fp16 checkpoints): the oracle
    replays them to the reference's fp16 pool contents bit for bit, and to its fp16 SDPA outputs (bitwise on the generating
    host, otherwise within the half-precision tolerance of the fp32 anchor)."""
    fr = FlowReplay(path, torch.float16)
    for ph in fr.phases():
        assert fnv64(bits16(ph.k_pool)) == int(fr.golden(ph.tag, "f16_k_pool_hash")), (ph.tag, "k pool")
        assert fnv64(bits16(ph.v_pool)) == int(fr.golden(ph.tag, "f16_v_pool_hash")), (ph.tag, "v pool")
        if fnv64(bits16(ph.out)) != int(fr.golden(ph.tag, "f16_o_hash")):
            warnings.warn(f"{fixture_id(path)}:{ph.tag}: fp16 SDPA differs bitwise from the generating host")
        rows, ref = fr.golden_o_f32(ph.tag)
        np.testing.assert_allclose(ph.out.float().numpy()[rows], ref, rtol=BF16_RTOL, atol=BF16_ATOL, err_msg=ph.tag)


def test_f16_fixture_present():
    assert len(_f16_fixtures()) >= 1


def test_out_cache_loc_dtype_promotion():
    table = np.arange(12, dtype=np.int32).reshape(3, 4)
    loc = M.out_cache_loc_ref(table, np.array([2, 0]), np.array([5, 3]), 2)
    assert loc.dtype == np.int64 and loc.tolist() == [(8 + 2) * 2 + 1, 1 * 2 + 1]


def test_page_pool_fifo_and_exhaustion():
    pool = M.RefPagePool(4, 16)
    assert pool.pages_for_prefill([17, 1], [0, 0]) == [[0, 1], [2]]
    assert pool.take(2) is None                      # kvcache.py:60-61: short -> None, nothing taken
    pool.give_back([1])
    assert pool.free_pages == [3, 1]
    assert pool.pages_for_decode([17, 18, 33]) == [[3], [], [1]]


def _load_traces():
    with open(os.path.join(os.path.dirname(__file__), "golden", "radix_traces.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("name", sorted(_load_traces().keys()))
def test_radix_trace(name):
    tr = _load_traces()[name]
    ps, prompts = tr["ps"], tr["prompts"]
    ctx = (max(len(p) for p in prompts) + 8) // ps * ps + ps
    pool = M.RefPagePool(tr["n_pages"], ps)
    pt = M.RefPageTable(ps, len(prompts) + 1, ctx)
    tree = R.RefRadixCache(pt, pool, ps)
    steps = iter(tr["steps"])

    def check(op, **extra):
        g = next(steps)
        assert g["op"] == op
        assert pool.free_pages == g["free_pages"], (op, extra)
        assert (tree.evictable, tree.protected, tree.total_size()) == (g["evictable"], g["protected"], g["total"]), op
        for k, v in extra.items():
            assert g[k] == v, (op, k, g[k], v)

    reqs, nodes = [], []
    for i, p in enumerate(prompts):
        rq = M.RefReq(fill_ids=list(p))
        rq.prefix_ppns, node = tree.match_prefix(rq.fill_ids[:max(len(p) - 1, 0)])
        meta = M.prepare_for_prefill_ref([rq], pool, pt)
        used = -(-len(p) // ps)
        check("prefill", req=i, prefix_ppns=rq.prefix_ppns, prefix_len=int(meta.prefix_lens[0]), row=rq.page_table_id,
              page_row=pt.table[rq.page_table_id, :used].tolist(), out_cache_loc=meta.out_cache_loc.tolist())
        node = tree.cache_unfinished_req(rq, node)
        check("cache_unfinished", req=i, prefix_ppns=rq.prefix_ppns, page_row=pt.table[rq.page_table_id, :used].tolist())
        reqs.append(rq)
        nodes.append(node)
    for i in tr["finish_order"]:
        toks = (list(prompts[i]) + [7000 + i, 7100 + i])[:-1]
        tree.cache_finished_req(reqs[i], toks, nodes[i])
        check("cache_finished", req=i, free_rows=list(pt.free_rows))
    for npg in tr["evict_pages"]:
        tree.evict(npg)
        check("evict", num_pages=npg)
    probe = list(prompts[0]) + [1, 2, 3]
    val, _ = tree.match_prefix(probe)
    check("match", key=probe, ppns=val)


def test_radix_repeated_pattern_quirk():
    """Reference radix_cache.py:263-267 keeps walking after a split without consuming the key."""
    tree = R.RefRadixCache(None, None, 1)
    tree.insert([1, 2, 1, 2], [10, 11, 12, 13])
    val, _ = tree.match_prefix([1, 2, 3])
    assert val == [10, 11, 12, 13]          # over-reports: only [1, 2] is really shared


def test_rope_ref_rotates_pairs():
    D, T = 8, 3
    inv = 1.0 / (10000 ** (torch.arange(0, D, 2, dtype=torch.float) / D))
    fr = torch.einsum("i,j->ij", torch.arange(16, dtype=torch.float), inv)
    cache = torch.cat((fr.cos(), fr.sin()), dim=-1)
    q = torch.randn(T, 2 * D)
    k = torch.randn(T, D)
    pos = torch.tensor([0, 5, 9])
    q2, k2 = A.rope_neox_ref(pos, q, k, D, cache)
    assert torch.allclose(q2[0], q[0]) and torch.allclose(k2[0], k[0])      # position 0 = identity
    assert torch.allclose(q2.view(T, 2, D).norm(dim=-1), q.view(T, 2, D).norm(dim=-1), atol=1e-5)
