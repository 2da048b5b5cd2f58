"""The N>1 path on CPU: two gloo ranks shard the attention path by heads the way bench.py --gpus N and the
reference do: identical host plans on every rank, per-rank KV pools with the local kv-heads
only, no collective inside attention, one SUM all-reduce after o_proj.  The device kernels cannot run here, so
the per-rank attention is the oracle's (allowed in tests/ only); everything else is product code."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, results):
    try:
        for p in (REPO, os.path.join(REPO, "mini-sglang_amd"), HERE, os.path.join(HERE, "golden")):
            if p not in sys.path:
                sys.path.insert(0, p)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        torch.set_num_threads(2)
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")      # one node: gloo must not go looking up the host name (it may not resolve)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import mi355_attn as m
        from mi355_attn.batch import plan_decode, plan_prefill
        from mi355_attn.tp import all_reduce_rows, row_parallel_linear, shard_heads
        from oracle import attention_ref as A
        from oracle import memory_ref as M

        HQ, HKV, D, ps, hidden = 8, 4, 32, 4, 48
        sh = shard_heads(HQ, HKV, D, world, rank)
        assert (sh.num_heads, sh.num_kv_heads, sh.q_head_start, sh.kv_head_start) == (HQ // world, HKV // world,
                                                                                     rank * HQ // world, rank * HKV // world)
        lens = [13, 4, 22]
        n_pages = 32
        kv = m.KVCache(size=n_pages * ps, page_size=ps, head_num=sh.num_kv_heads, head_dim=D, dtype=torch.float32,
                       layer_num=1, device="cpu")
        pm = m.PageManager(page_size=ps, max_req_num=4, max_context_len=64, device="cpu")
        reqs = [m.Req(i, list(range(100 * i, 100 * i + l))) for i, l in enumerate(lens)]
        for r in reqs:
            r.fill_ids = r.origin_input_ids
        plan = plan_prefill(reqs, pm, kv)
        # every rank must have derived the same plan without talking to the others (scheduler.py:224-227)
        mine = [plan.rows, plan.seq_lens, plan.prefix_lens, plan.pages.tolist(), plan.cu_q.tolist()]
        everyone = [None] * world
        dist.all_gather_object(everyone, mine)
        assert all(e == everyone[0] for e in everyone)
        for r, lo, hi in zip(plan.rows, plan.page_off[:-1], plan.page_off[1:]):
            pm.write_ppns_prefill(r, plan.pages[lo:hi].tolist())
        table = pm.page_table
        rows_t = torch.tensor(plan.rows)

        g = torch.Generator().manual_seed(7)                       # same full tensors on both ranks
        T = plan.T
        q = torch.randn((T, HQ * D), generator=g)
        k = torch.randn((T, HKV * D), generator=g)
        v = torch.randn((T, HKV * D), generator=g)
        w_o = torch.randn((HQ * D, hidden), generator=g) / 16
        row_of_token = np.repeat(np.arange(len(lens)), lens)
        pos = np.concatenate([np.arange(l) for l in lens])
        loc = torch.from_numpy(M.out_cache_loc_ref(table.numpy(), np.asarray(plan.rows)[row_of_token], pos, ps))

        def attend(q_, k_, v_, hq, hkv, kpool, vpool, seq, pre, decode):
            A.write_kv_ref(kpool, vpool, loc_cur, k_.reshape(-1, hkv, D), v_.reshape(-1, hkv, D))
            qq = q_.reshape(-1, hq, D)
            if decode:
                return A.decode_attn_ref(qq, kpool, vpool, table, rows_t, torch.tensor(seq), ps).reshape(-1, hq * D)
            return A.extend_attn_ref(qq, kpool, vpool, table, rows_t, torch.tensor(seq), torch.tensor(pre),
                                     ps).reshape(-1, hq * D)

        full_k = torch.zeros((n_pages * ps, HKV, D))
        full_v = torch.zeros_like(full_k)
        kp, vp = kv.get_kv_cache(0)
        loc_cur = loc
        for step in range(3):                                      # prefill, then two decode steps
            decode = step > 0
            if decode:
                lens, new_pages = plan_decode(lens, kv)
                for b, r in enumerate(plan.rows):
                    if new_pages[b] >= 0:
                        pm.write_ppns_decode(r, [(lens[b] - 1) // ps], [int(new_pages[b])])
                gathered = [None] * world
                dist.all_gather_object(gathered, new_pages.tolist())
                assert all(x == gathered[0] for x in gathered)
                pos = np.asarray(lens) - 1
                loc_cur = torch.from_numpy(M.out_cache_loc_ref(table.numpy(), np.asarray(plan.rows), pos, ps))
                q = torch.randn((len(lens), HQ * D), generator=g)
                k = torch.randn((len(lens), HKV * D), generator=g)
                v = torch.randn((len(lens), HKV * D), generator=g)
            pre = [0] * len(lens)
            o_local = attend(q[:, sh.q_columns()], k[:, sh.kv_columns()], v[:, sh.kv_columns()], sh.num_heads,
                             sh.num_kv_heads, kp, vp, lens, pre, decode)
            partial = o_local @ w_o[sh.q_columns()]               # row-parallel o_proj shard
            total = all_reduce_rows(partial.clone())
            o_full = attend(q, k, v, HQ, HKV, full_k, full_v, lens, pre, decode)
            want = o_full @ w_o
            assert torch.allclose(o_full[:, sh.q_columns()], o_local, atol=1e-6), "head shard differs from full run"
            assert torch.allclose(total, want, atol=1e-4, rtol=1e-4), float((total - want).abs().max())
            # f3: the same product through the pipelined o_proj (row blocks reduced while the next block multiplies)
            w_shard = w_o[sh.q_columns()].t().contiguous()        # [hidden, local Hq*D], the RowParallelLinear weight
            for chunks in (1, 3, 64):
                piped = row_parallel_linear(o_local, w_shard, chunks=chunks)
                assert piped.shape == want.shape
                assert torch.allclose(piped, want, atol=1e-4, rtol=1e-4), (chunks, float((piped - want).abs().max()))
            # the local pool holds exactly this rank's kv-head slice of the full pool
            assert torch.equal(kp, full_k[:, sh.kv_head_start: sh.kv_head_start + sh.num_kv_heads])
        dist.barrier()
        dist.destroy_process_group()
        status = "ok"
    except Exception as e:      # noqa: BLE001 - report to the parent
        import traceback
        status = traceback.format_exc() + repr(e)
    results.put((rank, status))


@pytest.mark.timeout(180)
def test_tp2_head_sharding_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"


def test_shard_heads_rejects_bad_splits():
    sys.path.insert(0, os.path.join(REPO, "mini-sglang_amd"))
    from mi355_attn.tp import shard_heads
    s = shard_heads(64, 8, 128, 8, 3)
    assert (s.num_heads, s.num_kv_heads, s.group_size, s.q_head_start, s.kv_head_start) == (8, 1, 8, 24, 3)
    assert s.q_columns() == slice(24 * 128, 32 * 128) and s.kv_columns() == slice(3 * 128, 4 * 128)
    with pytest.raises(ValueError):
        shard_heads(32, 8, 64, 16, 0)          # tp > kv heads: not a reference configuration
    with pytest.raises(ValueError):
        shard_heads(32, 8, 64, 2, 2)
