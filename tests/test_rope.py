"""Rotary embedding (+ fused KV slot write), SURVEY.md §8 f1.  Fixtures: tests/golden/rope.npz, written by
tests/golden/gen_golden_rope.py from the reference's own cache construction and in-tree formula."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from synth import bits16, synth  # noqa: E402

from oracle import attention_ref as A  # noqa: E402

CASES = ["llama32_1b", "llama3_8b", "llama2_like"]


def _load(name):
    z = np.load(os.path.join(HERE, "golden", "rope.npz"))
    D, max_pos, Hq, Hkv = (int(x) for x in z[f"{name}/meta"])
    sc = z[f"{name}/scaling"]
    scaling = None
    if sc.size:
        scaling = {"rope_type": "llama3", "factor": float(sc[0]), "low_freq_factor": float(sc[1]),
                   "high_freq_factor": float(sc[2]), "original_max_position_embeddings": int(sc[3])}
    return z, D, max_pos, Hq, Hkv, float(z[f"{name}/base"][0]), scaling


def _inputs(T, Hq, Hkv, D):
    return (synth((T, Hq * D), 11, amp=2.0, dtype=torch.bfloat16), synth((T, Hkv * D), 12, amp=2.0, dtype=torch.bfloat16))


@pytest.mark.parametrize("name", CASES)
def test_oracle_and_host_cache_match_reference_vectors(name):
    from mi355_attn.rotary import compute_cos_sin_cache
    z, D, max_pos, Hq, Hkv, base, scaling = _load(name)
    pos = torch.from_numpy(z[f"{name}/positions"])
    for cache in (A.rope_cache_ref(D, max_pos, base, scaling), compute_cos_sin_cache(D, max_pos, base, scaling)):
        assert cache.dtype == torch.float32 and tuple(cache.shape) == (max_pos, D)
        assert np.array_equal(cache[pos].numpy(), z[f"{name}/cache_rows"])
    q, k = _inputs(len(pos), Hq, Hkv, D)
    q_o, k_o = A.rope_neox_ref(pos, q, k, D, A.rope_cache_ref(D, max_pos, base, scaling))
    assert np.array_equal(bits16(q_o), z[f"{name}/q_out_bits"]) and np.array_equal(bits16(k_o), z[f"{name}/k_out_bits"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_rope_bit_exact_on_reference_vectors(name):
    from mi355_attn.rotary import RotaryEmbedding
    z, D, max_pos, Hq, Hkv, base, scaling = _load(name)
    pos = torch.from_numpy(z[f"{name}/positions"]).cuda()
    q, k = (t.cuda() for t in _inputs(len(pos), Hq, Hkv, D))
    rope = RotaryEmbedding(D, D, max_pos, base, scaling, device="cuda")
    q2, k2 = rope(pos, q, k)
    assert q2.data_ptr() == q.data_ptr()                       # in place, like the reference's kernel
    assert np.array_equal(bits16(q2), z[f"{name}/q_out_bits"]) and np.array_equal(bits16(k2), z[f"{name}/k_out_bits"])


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("Hq,Hkv,D,T", [(32, 8, 64, 300), (32, 8, 128, 77), (8, 1, 128, 1), (4, 4, 64, 513)])
def test_hip_rope_fused_store_matches_oracle(Hq, Hkv, D, T, dtype):
    from mi355_attn import ops
    max_pos, S = 4096, 2048
    cache = A.rope_cache_ref(D, max_pos, 500000.0)
    g = torch.Generator().manual_seed(T)
    qkv = torch.randn((T, (Hq + 2 * Hkv) * D), generator=g).to(dtype)                # packed projection output
    pos = torch.randint(0, max_pos, (T,), generator=g)
    loc = torch.randperm(S, generator=g)[:T]
    q, k, v = qkv.split([Hq * D, Hkv * D, Hkv * D], dim=-1)                          # strided views
    q_want, k_want = A.rope_neox_ref(pos, q, k, D, cache)
    kp = torch.randn((S, Hkv, D), generator=g).to(dtype)
    vp = torch.randn((S, Hkv, D), generator=g).to(dtype)
    kp_want, vp_want = kp.clone(), vp.clone()
    A.write_kv_ref(kp_want, vp_want, loc, k_want.view(T, Hkv, D), v.reshape(T, Hkv, D))
    d_qkv = qkv.cuda()
    dq, dk, dv = d_qkv.split([Hq * D, Hkv * D, Hkv * D], dim=-1)
    dkp, dvp = kp.cuda(), vp.cuda()
    ops.rope_kv_store(pos.cuda(), dq, dk, D, cache.cuda(), dv, dkp, dvp, loc.cuda())
    assert torch.equal(dq.cpu(), q_want) and torch.equal(dk.cpu(), k_want)
    assert torch.equal(dv.cpu(), v)                                                  # v untouched
    assert torch.equal(dkp.cpu(), kp_want) and torch.equal(dvp.cpu(), vp_want)
    # rope only (no pools): same rotation, pools untouched
    d2 = qkv.cuda()
    q2, k2, _ = d2.split([Hq * D, Hkv * D, Hkv * D], dim=-1)
    ops.rope_kv_store(pos.cuda(), q2, k2, D, cache.cuda())
    assert torch.equal(q2.cpu(), q_want) and torch.equal(k2.cpu(), k_want)


@pytest.mark.gpu
def test_hip_rope_rejects_bad_arguments():
    from mi355_attn import ops
    from mi355_attn._lib import MI355AttnError
    cache = torch.zeros((16, 64), device="cuda")
    q = torch.zeros((2, 128), dtype=torch.bfloat16, device="cuda")
    k = torch.zeros((2, 64), dtype=torch.bfloat16, device="cuda")
    with pytest.raises(MI355AttnError):
        ops.rope_kv_store(torch.zeros(2, dtype=torch.int32, device="cuda"), q, k, 64, cache)     # positions must be int64
    with pytest.raises(MI355AttnError):
        ops.rope_kv_store(torch.zeros(2, dtype=torch.int64, device="cuda"), q.float(), k.float(), 64, cache)
    ops.rope_kv_store(torch.zeros(0, dtype=torch.int64, device="cuda"), q[:0], k[:0], 64, cache)     # empty batch is fine
