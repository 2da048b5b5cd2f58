"""CPU: the product's HOST logic (no device kernels): page/row allocation plans, the radix tree, the C-ABI
surface, and loud failure when a device operation is attempted without a GPU."""
import json
import os
import re
import subprocess

import numpy as np
import pytest
import torch

from flow_util import FlowReplay, fixture_id, flow_fixtures
from oracle import memory_ref as M

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _m():
    import mi355_attn
    return mi355_attn


# ------------------------------------------------------------------------------------------------
# C ABI surface
# ------------------------------------------------------------------------------------------------
def _declared_symbols():
    text = ""
    for header in ("mi355attn.h", "mi355tp.h"):                     # both are served by libmi355attn.so
        with open(os.path.join(REPO, "include", header)) as f:
            text += re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(m355_[a-z0-9_]+)\s*\(", text)))


def _declared_arities():
    """name -> number of parameters, from the prototypes in include/*.h"""
    text = ""
    for header in ("mi355attn.h", "mi355tp.h"):
        with open(os.path.join(REPO, "include", header)) as f:
            text += re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    out = {}
    for name, args in re.findall(r"\b(m355_[a-z0-9_]+)\s*\(([^()]*)\)\s*;", text):
        args = args.strip()
        out[name] = 0 if args in ("", "void") else args.count(",") + 1
    return out


def test_library_exports_every_declared_symbol():
    from mi355_attn import _lib
    declared = _declared_symbols()
    assert len(declared) >= 10
    lib = _lib.load()                                   # raises if the .so or a bound symbol is missing
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/mi355attn.h but not exported"
    assert sorted(_lib.SIGNATURES) == declared, "ctypes table and header disagree"
    arity = _declared_arities()
    assert sorted(arity) == declared
    for name, (_, argtypes) in _lib.SIGNATURES.items():             # an extra or missing argument is a silent stack shift in ctypes
        assert len(argtypes) == arity[name], f"{name}: header has {arity[name]} parameters, the ctypes table {len(argtypes)}"
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r"\b(m355[a-z]*_[a-z0-9_]+)\b", out))
    assert exported == set(declared), "the product library exports exactly what include/*.h declares (no dev symbols)"
    assert lib.m355_version() == 310
    assert b"workspace" in lib.m355_strerror(-3)


def test_device_ops_refuse_cpu_tensors():
    m = _m()
    from mi355_attn import ops
    kv = m.KVCache(size=32, page_size=4, head_num=2, head_dim=64, dtype=torch.bfloat16, layer_num=1, device="cpu")
    with pytest.raises(RuntimeError, match="GPU"):
        kv.write_kv(0, torch.zeros(1, dtype=torch.int64), torch.zeros(1, 2, 64, dtype=torch.bfloat16),
                    torch.zeros(1, 2, 64, dtype=torch.bfloat16))
    with pytest.raises(RuntimeError, match="GPU"):
        ops.out_cache_loc(torch.zeros((2, 2), dtype=torch.int32), torch.zeros(1, dtype=torch.int64),
                          torch.zeros(1, dtype=torch.int64), 4)


# ------------------------------------------------------------------------------------------------
# allocation plans against the reference's page tables
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("path", flow_fixtures(), ids=fixture_id)
def test_plans_reproduce_reference_page_tables(path):
    m = _m()
    from mi355_attn.batch import plan_decode, plan_prefill
    fr = FlowReplay(path, torch.bfloat16)
    ps = fr.ps
    kv = m.KVCache(size=fr.S_pages * ps, page_size=ps, head_num=1, head_dim=8, dtype=torch.bfloat16, layer_num=1, device="cpu")
    pm = m.PageManager(page_size=ps, max_req_num=fr.max_req, max_context_len=fr.ctx, device="cpu")
    table = np.zeros((fr.max_req, fr.ctx // ps), dtype=np.int32)     # what the metadata kernel would write
    reqs, lens, rows = None, None, None
    for ph in fr.phases_meta_only():
        if not ph.decode:
            if ph.tag == "p1":
                reqs = [m.Req(i, fr.ids[i][: fr.cuts[i]]) for i in range(fr.n)]
                for r in reqs:
                    r.fill_ids = list(r.origin_input_ids)
            else:
                for i, r in enumerate(reqs):
                    r.fill_ids = list(fr.ids[i])
                    r.prefix_ppns = [int(x) for x in table[i, : fr.cuts[i] // ps]]
            plan = plan_prefill(reqs, pm, kv)
            rows, lens = plan.rows, plan.seq_lens
            assert np.array_equal(plan.input_ids, fr.golden(ph.tag, "input_ids"))
            assert plan.rows == fr.golden(ph.tag, "page_table_ids").tolist()
            assert plan.seq_lens == fr.golden(ph.tag, "seq_lens").tolist()
            assert plan.prefix_lens == fr.golden(ph.tag, "prefix_lens").tolist()
            assert [r.page_table_id for r in reqs] == plan.rows
            # emulate m355_prefill_prepare on the host to check the plan's content end to end
            pos, rot = [], []
            for b, row in enumerate(plan.rows):
                pg = plan.pages[plan.page_off[b]: plan.page_off[b + 1]]
                assert len(pg) == -(-plan.seq_lens[b] // ps)
                table[row, : len(pg)] = pg
                pos += list(range(plan.prefix_lens[b], plan.seq_lens[b]))
                rot += [row] * (plan.seq_lens[b] - plan.prefix_lens[b])
            assert plan.cu_q.tolist() == np.concatenate([[0], np.cumsum(np.array(plan.seq_lens) - np.array(plan.prefix_lens))]).tolist()
            packed = plan.packed_i32()
            assert packed.dtype == np.int32 and packed.size == (fr.n + 1) * 5 + len(plan.pages)
        else:
            lens, new_pages = plan_decode(lens, kv)
            pos = [l - 1 for l in lens]
            rot = rows
            for b, row in enumerate(rows):
                if new_pages[b] >= 0:
                    table[row, pos[b] // ps] = new_pages[b]
            assert lens == fr.golden(ph.tag, "seq_lens").tolist()
        assert np.array_equal(np.asarray(pos), fr.golden(ph.tag, "positions"))
        loc = M.out_cache_loc_ref(table, np.asarray(rot), np.asarray(pos), ps)
        assert np.array_equal(loc, fr.golden(ph.tag, "out_cache_loc")), ph.tag
        used = -(-max(lens) // ps)
        assert np.array_equal(table[np.asarray(rows), :used], fr.golden(ph.tag, "page_rows")), ph.tag
        assert kv.free_pages[:4] == [int(x) for x in fr.golden(ph.tag, "free_head")]


def test_pool_exhaustion_is_an_error_not_a_none():
    m = _m()
    from mi355_attn.batch import plan_prefill
    kv = m.KVCache(size=8, page_size=4, head_num=1, head_dim=8, dtype=torch.bfloat16, layer_num=1, device="cpu")
    pm = m.PageManager(page_size=4, max_req_num=2, max_context_len=64, device="cpu")
    r = m.Req(0, list(range(9)))
    r.fill_ids = r.origin_input_ids
    with pytest.raises(RuntimeError, match="exhausted"):
        plan_prefill([r], pm, kv)
    assert kv._allocate_pages(3) is None and kv.available_page_num() == 2     # kvcache.py:60-61 behaviour kept


def test_planner_refuses_requests_wider_than_the_page_table():
    """ceil(seq_len / ps) > max_page_num must fail on the host, before any row or page is taken (the
    reference fails on the row write, memory/page_manager.py:46, :54)."""
    m = _m()
    from mi355_attn.batch import plan_decode, plan_prefill
    ps, W = 4, 3
    kv = m.KVCache(size=64 * ps, page_size=ps, head_num=1, head_dim=8, dtype=torch.bfloat16, layer_num=1, device="cpu")
    pm = m.PageManager(page_size=ps, max_req_num=4, max_context_len=W * ps, device="cpu")
    ok, bad = m.Req(0, list(range(W * ps))), m.Req(1, list(range(W * ps + 1)))
    ok.fill_ids, bad.fill_ids = ok.origin_input_ids, bad.origin_input_ids
    with pytest.raises(RuntimeError, match="columns"):
        plan_prefill([ok, bad], pm, kv)
    assert kv.available_page_num() == 64 and pm.free_slots == [0, 1, 2, 3]          # nothing leaked
    plan = plan_prefill([ok], pm, kv)
    assert len(plan.pages) == W
    with pytest.raises(RuntimeError, match="columns"):
        plan_decode(plan.seq_lens, kv, pm)
    assert kv.available_page_num() == 64 - W
    # a pool that is too small is noticed before the rows are taken, too
    small = m.KVCache(size=2 * ps, page_size=ps, head_num=1, head_dim=8, dtype=torch.bfloat16, layer_num=1, device="cpu")
    pm2 = m.PageManager(page_size=ps, max_req_num=2, max_context_len=64, device="cpu")
    r = m.Req(2, list(range(9)))
    r.fill_ids = r.origin_input_ids
    with pytest.raises(RuntimeError, match="exhausted"):
        plan_prefill([r], pm2, small)
    assert pm2.free_slots == [0, 1] and small.available_page_num() == 2


def test_retract_returns_the_partly_filled_last_page():
    """retract_decode must give back ceil(len / ps) - prefix pages, not floor (engine/batch.py:428-431
    leaks the last page of every retracted request whose length is not a page multiple)."""
    m = _m()
    ps = 4
    kv = m.KVCache(size=16 * ps, page_size=ps, head_num=1, head_dim=8, dtype=torch.bfloat16, layer_num=1, device="cpu")
    pm = m.PageManager(page_size=ps, max_req_num=4, max_context_len=64, device="cpu")
    reqs = [m.Req(i, list(range(n))) for i, n in enumerate((9, 6, 13))]
    b = m.Batch(reqs, page_manager=pm, kvcache=kv, device="cpu")
    rows = pm.allocate(3)
    pages = kv.allocate_pages_prefill([9, 6, 13], [0, 0, 0])
    for r, row, pg in zip(reqs, rows, pages):
        r.page_table_id = row
        pm.write_ppns_prefill(row, pg)
    b.page_table_ids = torch.tensor(rows)
    b.seq_lens = torch.tensor([9, 6, 13])
    b.seq_lens_host = [9, 6, 13]
    assert kv.available_page_num() == 16 - (3 + 2 + 4)
    retracted, _ = b.retract_decode(retract_decode_steps=10 ** 6)       # cannot be satisfied: all but one go back
    assert len(retracted) == 2 and len(b.reqs) == 1
    kept = -(-b.seq_lens_host[0] // ps)
    assert kv.available_page_num() == 16 - kept                         # every page of the retracted requests is back
    assert sorted(pm.free_slots) == sorted(set(range(4)) - {b.reqs[0].page_table_id})


def test_vectorised_decode_plan_equals_the_per_request_allocator():
    """plan_decode (one numpy pass) hands out exactly the pages, in exactly the order, of the reference-shaped
    per-request loop allocate_pages_decode (memory/kvcache.py:49-56), for any page size; a short pool raises."""
    m = _m()
    from mi355_attn.batch import plan_decode
    rng = np.random.default_rng(5)
    for ps in (1, 2, 4, 16, 256):
        for _ in range(20):
            B = int(rng.integers(1, 70))
            lens = [int(x) for x in rng.integers(1, 5 * ps + 3, size=B)]
            size = int(rng.integers(1, 3 * B + 2)) * ps
            a = m.KVCache(size=size, page_size=ps, head_num=1, head_dim=8, dtype=torch.bfloat16, layer_num=1, device="cpu")
            b = m.KVCache(size=size, page_size=ps, head_num=1, head_dim=8, dtype=torch.bfloat16, layer_num=1, device="cpu")
            want = b.allocate_pages_decode([l + 1 for l in lens])
            if any(w is None for w in want):
                with pytest.raises(RuntimeError, match="exhausted"):
                    plan_decode(lens, a)
                continue
            new_lens, pages = plan_decode(lens, a)
            assert new_lens == [l + 1 for l in lens] and pages.dtype == np.int32
            assert pages.tolist() == [w[0] if w else -1 for w in want]
            assert a.free_pages == b.free_pages


# ------------------------------------------------------------------------------------------------
# radix tree against the reference traces
# ------------------------------------------------------------------------------------------------
def _traces():
    with open(os.path.join(REPO, "tests", "golden", "radix_traces.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("native", [False, True], ids=["python_tree", "native_tree"])
@pytest.mark.parametrize("name", sorted(_traces().keys()))
def test_radix_cache_trace(name, native):
    m = _m()
    from mi355_attn.batch import plan_prefill
    from mi355_attn.radix_cache import NativeRadixCache
    tr = _traces()[name]
    ps, prompts = tr["ps"], tr["prompts"]
    ctx = (max(len(p) for p in prompts) + 8) // ps * ps + ps
    kv = m.KVCache(size=tr["n_pages"] * ps, page_size=ps, head_num=1, head_dim=8, dtype=torch.bfloat16, layer_num=1, device="cpu")
    pm = m.PageManager(page_size=ps, max_req_num=len(prompts) + 1, max_context_len=ctx, device="cpu")
    tree = (NativeRadixCache if native else m.PagedRadixCache)(pm, kv, reference_split_walk=True)   # the reference's traces
    steps = iter(tr["steps"])

    def check(op, **extra):
        g = next(steps)
        assert g["op"] == op
        assert kv.free_pages == g["free_pages"], (op, extra)
        assert (tree.evictable_size(), tree.protected_size(), tree.total_size()) == (g["evictable"], g["protected"], g["total"]), op
        for k_, v_ in extra.items():
            assert g[k_] == v_, (op, k_, g[k_], v_)

    reqs = []
    for i, p in enumerate(prompts):
        r = m.Req(i, list(p), m.SamplingParams(max_new_tokens=4))
        r.init_next_round_input(tree)
        plan = plan_prefill([r], pm, kv)
        pg = plan.pages.tolist()
        pm.write_ppns_prefill(r.page_table_id, pg)                  # what m355_prefill_prepare does on the GPU
        pos = np.arange(plan.prefix_lens[0], plan.seq_lens[0])
        loc = M.out_cache_loc_ref(pm.page_table.numpy(), np.full(len(pos), r.page_table_id), pos, ps)
        used = -(-len(p) // ps)
        check("prefill", req=i, prefix_ppns=list(r.prefix_ppns), prefix_len=plan.prefix_lens[0], row=r.page_table_id,
              page_row=pm.page_table[r.page_table_id, :used].tolist(), out_cache_loc=loc.tolist())
        r.output_ids.append(7000 + i)
        tree.cache_unfinished_req(r)
        check("cache_unfinished", req=i, prefix_ppns=list(r.prefix_ppns), page_row=pm.page_table[r.page_table_id, :used].tolist())
        reqs.append(r)
    for i in tr["finish_order"]:
        reqs[i].output_ids.append(7100 + i)
        tree.cache_finished_req(reqs[i])
        check("cache_finished", req=i, free_rows=pm.free_slots)
    for npg in tr["evict_pages"]:
        tree.evict(npg)
        check("evict", num_pages=npg)
    probe = list(prompts[0]) + [1, 2, 3]
    val, _ = tree.match_prefix(probe)
    check("match", key=probe, ppns=val)


def test_native_radix_tree_equals_the_python_tree_on_random_traffic():
    """libmi355radix.so against mi355_attn.radix_cache.PagedRadixCache (itself pinned to the reference's traces) on
    random insert / match / lock / unlock / evict traffic over a small alphabet (many shared prefixes, partial-page
    matches, splits, the split-walk quirk at page_size 1): same pages, same sizes, same eviction order."""
    from mi355_attn.radix_cache import NativeRadixCache, PagedRadixCache

    class Pool:                      # records what the trees hand back
        def __init__(self):
            self.freed = []

        def _free_pages(self, pages):
            self.freed.extend(int(p) for p in pages)

    rng = np.random.default_rng(3)
    for ps in (1, 2, 4, 16):
        pools = [Pool(), Pool()]
        ref_walk = bool(ps % 4 == 1 or ps == 2)                     # both walks get random traffic
        trees = [PagedRadixCache(None, pools[0], page_size=ps, reference_split_walk=ref_walk),
                 NativeRadixCache(None, pools[1], page_size=ps, reference_split_walk=ref_walk)]
        held = []                                                    # (python node, native handle) pairs that are locked
        next_page = 0
        for step in range(400):
            op = rng.integers(0, 10)
            n_tok = int(rng.integers(0, 6 * ps + 3))
            key = [int(x) for x in rng.integers(0, 3, size=n_tok)]
            if op < 4:
                pages = list(range(next_page, next_page + n_tok // ps))
                next_page += len(pages)
                a, b = trees[0].insert(key, pages), trees[1].insert(key, pages)
                assert a == b, (ps, step, "insert")
            elif op < 7:
                try:
                    (pa, na), err = trees[0].match_prefix(key), None
                except Exception as e:                               # (never expected; keeps the comparison honest)
                    err = e
                assert err is None
                pb, nb = trees[1].match_prefix(key)
                assert pa == pb, (ps, step, "match", key)
                if rng.integers(0, 2) and len(held) < 6:
                    assert trees[0].inc_lock_ref(na) == trees[1].inc_lock_ref(nb)
                    held.append((na, nb))
            elif op < 8 and held:
                na, nb = held.pop(int(rng.integers(0, len(held))))
                assert trees[0].dec_lock_ref(na) == trees[1].dec_lock_ref(nb)
            else:
                k = int(rng.integers(1, 6))
                trees[0].evict(k)
                trees[1].evict(k)
                assert pools[0].freed == pools[1].freed, (ps, step, "evict order")
            sizes = [(t.evictable_size(), t.protected_size(), t.total_size()) for t in trees]
            assert sizes[0] == sizes[1], (ps, step, sizes)
        trees[0].evict(10 ** 6)
        trees[1].evict(10 ** 6)
        assert pools[0].freed == pools[1].freed


@pytest.mark.parametrize("native", [False, True], ids=["python_tree", "native_tree"])
def test_radix_match_never_returns_more_pages_than_the_key(native):
    """"abab" cached, "abc" queried at page_size 1.  The reference's walk after the split descends into the
    split-off tail and returns 4 pages for a 3-token key; the product's walk stops at the split."""
    from mi355_attn.radix_cache import NativeRadixCache, PagedRadixCache
    cls = NativeRadixCache if native else PagedRadixCache
    a, b, c = 11, 22, 33
    for ref_walk, want in ((False, [100, 101]), (True, [100, 101, 102, 103])):
        tree = cls(None, None, page_size=1, reference_split_walk=ref_walk)
        assert tree.insert([a, b, a, b], [100, 101, 102, 103]) == 0
        ppns, _ = tree.match_prefix([a, b, c])
        assert ppns == want
        if not ref_walk:
            assert len(ppns) <= 3
    rng = np.random.default_rng(0)
    for ps in (1, 2, 4):                                            # random traffic: a match is always a true prefix
        tree = cls(None, None, page_size=ps)
        stored, nxt = {}, 0
        for _ in range(300):
            key = [int(x) for x in rng.integers(0, 2, size=int(rng.integers(0, 7 * ps)))]
            if rng.integers(0, 2):
                pages = list(range(nxt, nxt + len(key) // ps))
                nxt += len(pages)
                got = tree.insert(key, pages) // ps
                for i in range(got, len(pages)):
                    stored[tuple(key[: (i + 1) * ps])] = pages[i]
            else:
                ppns, _ = tree.match_prefix(key)
                assert len(ppns) * ps <= len(key)
                for i, pg in enumerate(ppns):
                    assert stored[tuple(key[: (i + 1) * ps])] == pg


def test_native_radix_library_exports_its_header():
    import ctypes
    import re
    hdr = open(os.path.join(REPO, "include", "mi355radix.h")).read()
    names = sorted(set(re.findall(r"\b(m355r_[a-z_]+)\s*\(", hdr)))
    assert len(names) >= 12
    lib = ctypes.CDLL(os.path.join(REPO, "mini-sglang_amd", "lib", "libmi355radix.so"))
    for n in names:
        assert hasattr(lib, n), n
    lib.m355r_version.restype = ctypes.c_int
    assert lib.m355r_version() == 1



def test_group_sizes_run_in_the_next_built_one():
    """Backend: query heads per kv head outside {1, 2, 4, 8} are padded up (tests/test_gpu_configs.py checks the results);
    more than 8, or a head count that is no multiple of the kv heads, is refused."""
    from mi355_attn._lib import MI355AttnError
    from mi355_attn.backend import _group_pad
    assert [_group_pad(h, 8) for h in (8, 16, 24, 32, 40, 56, 64)] == [1, 2, 4, 4, 8, 8, 8]
    with pytest.raises(MI355AttnError):
        _group_pad(72, 8)
    with pytest.raises(MI355AttnError):
        _group_pad(30, 8)


# ------------------------------------------------------------------------------------------------
# backend selection by name (the flag engine/model_runner.py:90-91 and utils/args.py lack)
# ------------------------------------------------------------------------------------------------
def test_select_backend_dispatch_and_refusal(monkeypatch):
    import argparse
    import sys
    import types
    from types import SimpleNamespace
    m = _m()
    runner = SimpleNamespace(device="cuda", page_size=16)
    # this package's backend, under every spelling
    for name in ("mi355", "MI355", " mi355x ", "hip"):
        b = m.select_backend(name, runner)
        assert type(b) is m.MI355AttnBackend and b.page_size == 16 and b.device == "cuda"
    # unknown names are refused with the choices listed - no default, no fallback
    with pytest.raises(ValueError, match="mi355, torch_native, fa3"):
        m.select_backend("flashinfer", runner)
    # the reference's backends are imported lazily, by name, from whatever `minisglang` is on the path
    assert not any(k.startswith("minisglang") for k in sys.modules), "importing mi355_attn must not import the reference"
    made = []

    class FakeNative:
        def __init__(self, model_runner):
            made.append(model_runner)

    pkgs = ["minisglang", "minisglang.layers", "minisglang.layers.attention_backends"]
    for p in pkgs:
        mod = types.ModuleType(p)
        mod.__path__ = []                                       # a package with nothing importable inside
        monkeypatch.setitem(sys.modules, p, mod)
    leaf = types.ModuleType(pkgs[-1] + ".torch_native_backend")
    leaf.TorchNativeAttnBackend = FakeNative
    monkeypatch.setitem(sys.modules, leaf.__name__, leaf)
    for name in ("torch_native", "torch-native", "sdpa"):
        assert type(m.select_backend(name, runner)) is FakeNative
    assert made == [runner] * 3
    # ... and one whose module is missing (FA3 needs sgl_kernel, flash_attention_backend.py:5) says so, loudly
    with pytest.raises(ImportError, match="fa3"):
        m.select_backend("fa3", runner)
    # the argparse piece: --attention-backend next to --page-size
    ap = argparse.ArgumentParser()
    ap.add_argument("--page-size", type=int, default=1)
    m.add_backend_argument(ap)
    assert ap.parse_args([]).attention_backend == "mi355"
    assert ap.parse_args(["--attention-backend", "Torch-Native"]).attention_backend == "torch_native"
    with pytest.raises(SystemExit):
        ap.parse_args(["--attention-backend", "nope"])
