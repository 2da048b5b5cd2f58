"""bench_one_batch counterpart on tiny synthetic-weight Llama shells: the three-phase sequence
prefill(cut) -> extend over the cached prefix -> decode must agree with a single-shot prefill through the whole
model (prefill MFMA kernel with and without prefix, rope+store, metadata kernels, decode kernel)."""
import os
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "mini-sglang_amd"))

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("model,ps,cut", [("tiny", 16, 32), ("tiny", 1, 5), ("tiny128", 16, 48), ("tiny128", 4, 8)])
def test_three_phase_matches_single_shot(model, ps, cut):
    import bench_one_batch as bob
    from harness.llama_shell import SHAPES, HarnessRunner
    shape = SHAPES[model]
    B, L, out_len = 3, 90, 6
    runner = HarnessRunner(shape, ps, max_total_tokens=B * 2 * (L + out_len + 2 * ps) + 64 * ps, max_reqs=B,
                           context_len=L + out_len + ps)
    prompts = [p[:n] for p, n in zip(bob.synth_prompts(B, L, shape.vocab, seed=3), (90, 61, 77))]
    logits3, ids3 = bob.three_phase(runner, prompts, cut, out_len)
    bob.reset_pools(runner)
    logits1, ids1 = bob.three_phase(runner, prompts, 0, out_len)
    scale = float(logits1.float().abs().max())
    assert torch.isfinite(logits3).all() and scale > 0
    # (a smoke bar for the harness plumbing - cut + extend against a single shot through random weights; the parity
    # evidence for the kernels is tests/test_gpu_parity.py and tests/test_gpu_configs.py, against the pinned oracle)
    assert float((logits3.float() - logits1.float()).abs().max()) <= 0.05 * scale
    assert torch.equal(ids3[:, 0], ids1[:, 0]) or float((ids3 == ids1).float().mean()) > 0.8


def test_decode_step_agrees_with_prefill_of_longer_prompt():
    """Logits of a decode step == logits of a prefill that already contains the decoded token (decode and prefill
    kernels, rope positions and slot writes must all line up)."""
    import bench_one_batch as bob
    from harness.llama_shell import SHAPES, HarnessRunner
    shape, ps, B, L = SHAPES["tiny"], 16, 4, 50
    runner = HarnessRunner(shape, ps, max_total_tokens=B * (L + 40) + 64 * ps, max_reqs=B, context_len=L + 40)
    prompts = bob.synth_prompts(B, L, shape.vocab, seed=5)
    nxt, _, batch = bob.extend(bob.make_reqs(prompts), runner)
    nxt2, logits_dec = bob.decode(nxt, batch, runner)
    bob.reset_pools(runner)
    longer = [p + [int(t)] for p, t in zip(prompts, nxt.tolist())]
    _, logits_pre, _ = bob.extend(bob.make_reqs(longer), runner)
    scale = float(logits_pre.float().abs().max())
    assert float((logits_dec.float() - logits_pre.float()).abs().max()) <= 0.05 * scale


def test_latency_mode_emits_the_result_record(tmp_path):
    import bench_one_batch as bob
    from harness.llama_shell import SHAPES, HarnessRunner
    runner = HarnessRunner(SHAPES["tiny"], 16, max_total_tokens=4096, max_reqs=2, context_len=128)
    res = bob.latency_test(runner, bob.BenchArgs(model="tiny"), 2, 64, 4, lambda *a, **k: None, 1)
    assert res["batch_size"] == 2 and res["decode_throughput"] > 0 and res["prefill_latency"] > 0


@pytest.mark.parametrize("model,ps", [("tiny", 16), ("tiny128", 4)])
def test_graphed_decode_equals_the_eager_loop(model, ps):
    """One decode step captured as a HIP graph (device free list, in-place metadata, fixed plan layout) and replayed
    must generate exactly the eager loop's tokens, across page boundaries."""
    import bench_one_batch as bob
    from harness.llama_shell import SHAPES, HarnessRunner
    shape = SHAPES[model]
    B, L, out_len = 4, 40, 3 * ps + 4 if ps <= 4 else 24
    runner = HarnessRunner(shape, ps, max_total_tokens=B * (L + out_len + 2 * ps) + 64 * ps, max_reqs=B, context_len=L + out_len + ps,
                           device="cuda")
    prompts = [p[: L - 3 * i] for i, p in enumerate(bob.synth_prompts(B, L, shape.vocab, seed=5))]
    _, eager = bob.three_phase(runner, prompts, 0, out_len)
    bob.reset_pools(runner)
    _, graphed = bob.three_phase(runner, prompts, 0, out_len, graph=True)
    assert torch.equal(eager, graphed)
    assert runner.kvcache.ring.snapshot() == runner.kvcache.free_pages
    # while a graph lives the backend refuses to replace the buffers its kernels point into; released, it may again
    be = runner.attn_backend
    assert not be.buffers_pinned
    be.buffers_pinned = True
    with pytest.raises(RuntimeError, match="pins the workspace"):
        be._ws(be._workspace.numel() * 4 if be._workspace is not None else 1 << 20, "cuda")
    be.buffers_pinned = False


def test_three_phase_in_half_precision():
    """The same three-phase check with an IEEE-half model (Llama-2-style checkpoints): fp16 goes through the same
    matrix-core kernels (rope + store, prefill, planned decode with its merge)."""
    import bench_one_batch as bob
    from harness.llama_shell import SHAPES, HarnessRunner
    shape = SHAPES["tiny128"]
    B, L, out_len, ps, cut = 3, 90, 6, 16, 48
    runner = HarnessRunner(shape, ps, max_total_tokens=B * 2 * (L + out_len + 2 * ps) + 64 * ps, max_reqs=B,
                           context_len=L + out_len + ps, dtype=torch.float16)
    prompts = [p[:n] for p, n in zip(bob.synth_prompts(B, L, shape.vocab, seed=3), (90, 61, 77))]
    logits3, ids3 = bob.three_phase(runner, prompts, cut, out_len)
    bob.reset_pools(runner)
    logits1, ids1 = bob.three_phase(runner, prompts, 0, out_len)
    scale = float(logits1.float().abs().max())
    assert logits3.dtype == torch.float16 and torch.isfinite(logits3).all() and scale > 0
    assert float((logits3.float() - logits1.float()).abs().max()) <= 0.05 * scale
    assert torch.equal(ids3[:, 0], ids1[:, 0]) or float((ids3 == ids1).float().mean()) > 0.8
