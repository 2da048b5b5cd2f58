#!/usr/bin/env python3
"""Golden vectors for the rotary embedding, produced by the REFERENCE's own in-tree code on CPU.

    python tests/golden/gen_golden_rope.py          (build container only: needs /root/reference)

The reference module layers/rotary_embedding.py imports ``apply_rope_with_cos_sin_cache_inplace`` from the
CUDA-only ``sgl_kernel`` wheel at import time (:5).  A stub module is registered HERE ONLY so that the module
imports; the stubbed function is never called.  What is taken from the reference: ``get_rope`` ->
``RotaryEmbedding`` / ``Llama3RotaryEmbedding`` cache construction (:45-63, :116-137) and the in-tree formula
``_apply_rotary_emb`` (:8-18).  Parity with the sgl_kernel kernel itself stays UNPINNED.
The oracle (oracle/attention_ref.py: rope_cache_ref, rope_neox_ref) is asserted equal to both while writing.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True

_stub = types.ModuleType("sgl_kernel")
_stub.apply_rope_with_cos_sin_cache_inplace = None        # never called
sys.modules["sgl_kernel"] = _stub
sys.path.insert(0, "/root/reference")
from minisglang.layers import rotary_embedding as ref_rope  # noqa: E402

from synth import bits16, synth  # noqa: E402
from oracle import attention_ref as A  # noqa: E402

CASES = {
    # name: head_dim, max_position, base, rope_scaling, (Hq, Hkv), sampled positions
    "llama32_1b": (64, 131072, 500000.0,
                   {"rope_type": "llama3", "factor": 32.0, "low_freq_factor": 1.0, "high_freq_factor": 4.0,
                    "original_max_position_embeddings": 8192}, (4, 2),
                   [0, 1, 2, 17, 255, 4095, 8191, 8192, 65535, 131071]),
    "llama3_8b": (128, 8192, 500000.0, None, (4, 1), [0, 1, 3, 100, 1023, 1024, 8191]),
    "llama2_like": (64, 4096, 10000.0, None, (2, 2), [0, 5, 4095]),
}

out = {}
for name, (D, max_pos, base, scaling, (Hq, Hkv), pos) in CASES.items():
    rope = ref_rope.get_rope(head_size=D, rotary_dim=D, max_position=max_pos, base=base,
                             rope_scaling=dict(scaling) if scaling else None)
    cache = rope.cos_sin_cache
    mine = A.rope_cache_ref(D, max_pos, base, scaling)
    assert torch.equal(cache, mine), name
    positions = torch.tensor(pos, dtype=torch.int64)
    T = len(pos)
    q = synth((T, Hq * D), 11, amp=2.0, dtype=torch.bfloat16)
    k = synth((T, Hkv * D), 12, amp=2.0, dtype=torch.bfloat16)
    cs = cache[positions]
    cos, sin = cs.chunk(2, dim=-1)
    q_ref = ref_rope._apply_rotary_emb(q.view(T, Hq, D), cos, sin).reshape(T, Hq * D)
    k_ref = ref_rope._apply_rotary_emb(k.view(T, Hkv, D), cos, sin).reshape(T, Hkv * D)
    q_o, k_o = A.rope_neox_ref(positions, q, k, D, cache)
    assert torch.equal(q_o, q_ref) and torch.equal(k_o, k_ref), name
    out[f"{name}/positions"] = positions.numpy()
    out[f"{name}/cache_rows"] = cs.numpy()
    out[f"{name}/q_out_bits"] = bits16(q_ref)
    out[f"{name}/k_out_bits"] = bits16(k_ref)
    out[f"{name}/meta"] = np.array([D, max_pos, Hq, Hkv], dtype=np.int64)
    out[f"{name}/base"] = np.array([base], dtype=np.float64)
    out[f"{name}/scaling"] = np.array([scaling[k] for k in ("factor", "low_freq_factor", "high_freq_factor",
                                                            "original_max_position_embeddings")] if scaling else [],
                                      dtype=np.float64)
path = os.path.join(HERE, "rope.npz")
np.savez_compressed(path, **out)
print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")
