"""The allocation-free entry point m355_decode_store_attn_planned_static (include/mi355attn.h) against the workspace
form m355_decode_store_attn_planned, through the C ABI: same kernels, so pools AND outputs must be bit-identical
(the bar for byte work; the workspace form itself is held to the bf16 tolerance against the oracle elsewhere).
Calls in a row on the same GPU share the library's scratch array (used by the per-wave-range kernel's partials; the
single-launch share kernel needs none): a smaller call after a larger one must not see anything of it.  Collected last (test_zz_*): the round's end runs pytest -x."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _case(B, Hq, Hkv, D, ps, lens, seed):
    g = torch.Generator().manual_seed(seed)
    npages = sum(-(-l // ps) for l in lens) + 2
    perm = torch.randperm(npages, generator=g).to(torch.int32)
    W = max(-(-l // ps) for l in lens) + 1
    table = torch.zeros((B, W), dtype=torch.int32)
    off = 0
    for b, l in enumerate(lens):
        n = -(-l // ps)
        table[b, :n] = perm[off:off + n]
        off += n
    kp = torch.randn((npages * ps, Hkv, D), generator=g).to(torch.bfloat16)
    vp = torch.randn((npages * ps, Hkv, D), generator=g).to(torch.bfloat16)
    qkv = (torch.randn((B, (Hq + 2 * Hkv) * D), generator=g) * 2).to(torch.bfloat16)
    return table, kp, vp, qkv


def _dev_pool(pool, pad):
    """[S, Hkv, D] on the device with `pad` unused elements behind every token row (0: contiguous, the reference's layout)."""
    S, Hkv, D = pool.shape
    full = torch.zeros((S, Hkv * D + pad), dtype=pool.dtype, device=DEV)
    full[:, :Hkv * D] = pool.reshape(S, Hkv * D).to(DEV)
    return full[:, :Hkv * D].unflatten(1, (Hkv, D))


def _run_both(ops, lib, B, Hq, Hkv, D, ps, lens, seed, pad=0):
    from mi355_attn._lib import current_stream
    table, kp, vp, qkv = _case(B, Hq, Hkv, D, ps, lens, seed)
    pitch = Hkv * D + pad
    dqkv = qkv.to(DEV)
    q, k, v = dqkv.split([Hq * D, Hkv * D, Hkv * D], dim=-1)                 # strided views of the packed projection
    dt, dr, dl = table.to(DEV), torch.arange(B, device=DEV), torch.tensor(lens, device=DEV)
    cap = sum(lens)
    plan = ops.decode_plan(dt, dr, dl, ps, Hq, Hkv, D, cap, pool_row_stride=pitch)
    # workspace form
    kp1, vp1 = _dev_pool(kp, pad), _dev_pool(vp, pad)
    need = int(lib.m355_decode_planned_workspace_bytes(B, Hq, Hkv, D, cap, pitch))
    ws = torch.empty(need, dtype=torch.uint8, device=DEV)
    o1 = torch.full((B, Hq * D), 7.0, dtype=torch.bfloat16, device=DEV)
    rc = lib.m355_decode_store_attn_planned(q.data_ptr(), q.stride(0), k.data_ptr(), k.stride(0), v.data_ptr(), v.stride(0),
                                            kp1.data_ptr(), vp1.data_ptr(), pitch, plan.buf.data_ptr(), o1.data_ptr(), o1.stride(0),
                                            B, Hq, Hkv, D, D ** -0.5, cap, ws.data_ptr(), need, 0, current_stream())
    assert rc == 0
    # allocation-free form: every buffer it touches exists before the call
    kp2, vp2 = _dev_pool(kp, pad), _dev_pool(vp, pad)
    o2 = torch.full((B, Hq * D), -7.0, dtype=torch.bfloat16, device=DEV)
    torch.cuda.synchronize()
    before = torch.cuda.memory_allocated()
    rc = lib.m355_decode_store_attn_planned_static(q.data_ptr(), q.stride(0), k.data_ptr(), k.stride(0), v.data_ptr(),
                                                   v.stride(0), kp2.data_ptr(), vp2.data_ptr(), pitch, plan.buf.data_ptr(),
                                                   o2.data_ptr(), o2.stride(0), B, Hq, Hkv, D, D ** -0.5, cap, 0,
                                                   current_stream())
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated() == before
    assert torch.equal(kp1, kp2) and torch.equal(vp1, vp2)
    assert torch.equal(o1, o2)
    assert torch.isfinite(o2.float()).all()
    assert plan.error() == 0
    return need


def test_static_scratch_entry_equals_workspace_entry():
    import mi355_attn
    from mi355_attn import _lib, ops
    lib = _lib.load()
    rng = np.random.default_rng(0)
    # 1. cfg2-sized: B = 64 ragged, 1B head shape: served by the single-launch share kernel, which needs NO scratch (the
    #    partials of requests a share boundary cuts travel inside the plan buffer): need1 == 0, the array is not touched
    lens = [int(x) for x in rng.integers(64, 2049, size=64)]
    #    (rows padded as mi355_attn.KVCache pads them; on the reference's contiguous rows this shape goes to the range kernel: 1b)
    need1 = _run_both(ops, lib, 64, 32, 8, 64, 16, lens, seed=1, pad=64)
    need1b = _run_both(ops, lib, 64, 32, 8, 64, 16, lens, seed=1)
    assert 0 < need1b <= ops.STATIC_SCRATCH_BYTES
    # 2. right after it, a SMALLER batch with another head shape (few requests per CU: per-wave ranges + merge, whose
    #    fp32 partials DO live in the scratch array): need2 > 0 - this is the case that exercises the array
    lens2 = [int(x) for x in rng.integers(1, 700, size=5)]
    need2 = _run_both(ops, lib, 5, 8, 1, 128, 16, lens2, seed=2)
    assert need1 == 0 and need2 > 0, "case 2 must be the scratch-using kernel"
    assert sum(lens2) < sum(lens) and max(need1, need2) <= ops.STATIC_SCRATCH_BYTES
    # 2b. a larger scratch user first, then the smaller one again: the second call must see nothing of the first
    lens3 = [int(x) for x in rng.integers(300, 1500, size=12)]
    need3 = _run_both(ops, lib, 12, 8, 1, 128, 16, lens3, seed=4)
    assert need3 > need2
    _run_both(ops, lib, 5, 8, 1, 128, 16, lens2, seed=2)
    # 3. the wrapper form, page_size 1
    table, kp, vp, qkv = _case(3, 16, 4, 128, 1, [300, 17, 1], 3)
    dqkv = qkv.to(DEV)
    q, k, v = dqkv.split([16 * 128, 4 * 128, 4 * 128], dim=-1)
    dt, dr, dl = table.to(DEV), torch.arange(3, device=DEV), torch.tensor([300, 17, 1], device=DEV)
    plan = ops.decode_plan(dt, dr, dl, 1, 16, 4, 128, 318)
    a = ops.decode_attn_planned(q, kp.to(DEV), vp.to(DEV), plan, 16, k_new=k, v_new=v)
    out = torch.empty_like(a)
    b = ops.decode_attn_planned(q, kp.to(DEV), vp.to(DEV), plan, 16, k_new=k, v_new=v, static_scratch=True, out=out)
    assert b.data_ptr() == out.data_ptr() and torch.equal(a, b)


def test_static_scratch_entry_refuses_what_does_not_fit():
    from mi355_attn import _lib, ops
    lib = _lib.load()
    B, Hq, Hkv, D, cap = 16, 8, 1, 128, 1 << 21        # few, very long requests: thousands of fp32 partials
    assert lib.m355_decode_planned_workspace_bytes(B, Hq, Hkv, D, cap, 0) > ops.STATIC_SCRATCH_BYTES
    x = torch.zeros(4096, dtype=torch.bfloat16, device=DEV)       # any valid, aligned device pointer: nothing is launched
    rc = lib.m355_decode_store_attn_planned_static(x.data_ptr(), Hq * D, x.data_ptr(), Hkv * D, x.data_ptr(), Hkv * D,
                                                   x.data_ptr(), x.data_ptr(), 0, x.data_ptr(), x.data_ptr(), Hq * D, B, Hq,
                                                   Hkv, D, 0.1, cap, 0, None)
    assert rc == -5
    assert b"M355_STATIC_SCRATCH_BYTES" in lib.m355_strerror(rc)
    with pytest.raises(_lib.MI355AttnError, match="scratch"):
        _lib.check(rc, "m355_decode_store_attn_planned_static")
