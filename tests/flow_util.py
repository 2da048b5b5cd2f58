"""Replay of a golden "flow" fixture (prefill cut -> extend over cached prefix -> decode steps)
with the oracle only.  Shared by the CPU oracle-vs-golden test and by the GPU parity tests, which
run the HIP path on the same inputs phase by phase."""
from __future__ import annotations

import glob
import os
from dataclasses import dataclass
from typing import Iterator, List, Optional

import numpy as np
import torch

from synth import synth, synth_ints
from oracle import attention_ref as A
from oracle import memory_ref as M

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def flow_fixtures() -> List[str]:
    return sorted(glob.glob(os.path.join(GOLDEN, "flow_*.npz")))


def fixture_id(path: str) -> str:
    return os.path.basename(path)[len("flow_"):-len(".npz")]


def qkv_packed(T, Hq, Hkv, D, seed, dtype):
    return synth((T, (Hq + 2 * Hkv) * D), seed, amp=2.0, dtype=dtype)


def split_qkv(qkv, Hq, Hkv, D):
    """q, k, v as strided views of the packed projection output (reference models/llama.py:133-134)."""
    return qkv.split([Hq * D, Hkv * D, Hkv * D], dim=-1)


@dataclass
class Phase:
    tag: str                       # p1 | p2 | d<k>
    decode: bool
    meta: M.RefBatchMeta           # oracle metadata for this phase
    reqs: list                     # RefReq list (prefill phases) - fill_ids / prefix_ppns as the engine sees them
    next_ids: Optional[list]       # decode phases: the token ids fed in
    qkv: torch.Tensor              # [T, (Hq+2Hkv)*D] packed projection output
    q: torch.Tensor                # [T, Hq*D]   strided view of qkv
    k: torch.Tensor                # [T, Hkv*D]
    v: torch.Tensor
    k_pool: torch.Tensor           # oracle pools AFTER this phase's write
    v_pool: torch.Tensor
    table: np.ndarray              # oracle page table after this phase
    free_pages: list
    out: torch.Tensor              # oracle attention output [T, Hq*D] in `dtype`


class FlowReplay:
    def __init__(self, path: str, dtype=torch.bfloat16):
        self.g = np.load(path)
        g = self.g
        self.ps = int(g["ps"])
        self.Hq, self.Hkv, self.D = (int(x) for x in g["heads"])
        self.lens = [int(x) for x in g["lens"]]
        self.cuts = [int(x) for x in g["cuts"]]
        self.steps = int(g["steps"])
        self.ctx = int(g["ctx"])
        self.S_pages = int(g["S_pages"])
        self.n = len(self.lens)
        self.dtype = dtype
        self.ids = [synth_ints(l, 100 + i, 0, 31999) for i, l in enumerate(self.lens)]
        for i in range(self.n):
            assert np.array_equal(np.asarray(self.ids[i]), g[f"ids{i}"]), "synth_ints drifted from the fixture"
        self.max_req = 2 * self.n + 1

    def phases_meta_only(self) -> Iterator[Phase]:
        """Same walk without tensors: only tag / decode / meta / table / free_pages are filled in."""
        return self.phases(compute=False)

    def phases(self, compute: bool = True) -> Iterator[Phase]:
        ps, Hq, Hkv, D, n, dt = self.ps, self.Hq, self.Hkv, self.D, self.n, self.dtype
        pool = M.RefPagePool(self.S_pages, ps)
        pt = M.RefPageTable(ps, self.max_req, self.ctx)
        k_pool = torch.zeros((self.S_pages * ps, Hkv, D), dtype=dt)
        v_pool = torch.zeros_like(k_pool)

        def attend(meta, q, decode):
            args = (k_pool, v_pool, torch.from_numpy(pt.table), torch.from_numpy(meta.page_table_ids),
                    torch.from_numpy(meta.seq_lens))
            T = q.shape[0]
            if decode:
                return A.decode_attn_ref(q.reshape(T, Hq, D), *args, ps).reshape(T, Hq * D)
            return A.extend_attn_ref(q.reshape(T, Hq, D), *args, torch.from_numpy(meta.prefix_lens), ps).reshape(T, Hq * D)

        def step(tag, decode, meta, reqs, next_ids, seed):
            if not compute:
                return Phase(tag, decode, meta, reqs, next_ids, None, None, None, None, None, None, pt.table,
                             list(pool.free_pages), None)
            T = int(meta.input_ids.shape[0])
            qkv = qkv_packed(T, Hq, Hkv, D, seed, dt)
            q, k, v = split_qkv(qkv, Hq, Hkv, D)
            A.write_kv_ref(k_pool, v_pool, torch.from_numpy(meta.out_cache_loc), k.view(-1, Hkv, D), v.view(-1, Hkv, D))
            out = attend(meta, q, decode)
            return Phase(tag, decode, meta, reqs, next_ids, qkv, q, k, v, k_pool, v_pool, pt.table,
                         list(pool.free_pages), out)

        reqs = [M.RefReq(fill_ids=list(self.ids[i][:self.cuts[i]])) for i in range(n)]
        meta = M.prepare_for_prefill_ref(reqs, pool, pt)
        yield step("p1", False, meta, reqs, None, 1)

        reqs = [M.RefReq(fill_ids=list(self.ids[i]),
                         prefix_ppns=[int(x) for x in pt.table[i, : self.cuts[i] // ps]]) for i in range(n)]
        meta = M.prepare_for_prefill_ref(reqs, pool, pt)
        yield step("p2", False, meta, reqs, None, 2)

        for s in range(self.steps):
            nxt = synth_ints(n, 900 + s, 0, 31999)
            meta = M.prepare_for_decode_ref(meta, nxt, pool, pt)
            yield step(f"d{s}", True, meta, None, nxt, 10 + s)

    # ---- fixture accessors
    def golden(self, tag: str, field: str):
        return self.g[f"{tag}.{field}"]

    def golden_o_f32(self, tag: str):
        """(row indices, fp32 reference output on those rows)."""
        return self.g[f"{tag}.o_f32_rows"], self.g[f"{tag}.o_f32"]


# tolerance of a bf16-output attention kernel against the fp32 anchor (BASELINE.json north_star:
# "attention output within a stated bf16 tolerance against an fp32 reference").  Outputs are
# convex combinations of V rows with |v| <= 2, so one bf16 ulp at magnitude 2 is 2^-7 = 7.8e-3;
# inputs/P rounding add a little more.
BF16_ATOL = 2e-2
BF16_RTOL = 2e-2
