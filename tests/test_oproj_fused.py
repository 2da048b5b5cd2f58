"""GPU: the fused o_proj + all-reduce epilogue (include/mi355tp.h) against F.linear + SUM all-reduce
(reference layers/linear.py:228-232).

One GPU per box: every rank of the world lives on cuda:0 here (FusedRowParallel(local_world=N): all ranks' receive buffers
and flags on this device, all ranks run in one launch), so the GEMM, the slot/flag indexing, the epoch and buffer-half
logic and the rank-order reduction are exercised; what a single GPU cannot show is xGMI visibility of peer writes.
Bar: the reference reduces bf16 partials (F.linear rounds, then all_reduce adds), so the expected value is the fp32
sum of the bf16-rounded per-rank products, rounded once; the kernel adds the same rounded partials in fp32 in rank
order -> equal up to the GEMM's own accumulation order inside a partial: atol = rtol = 2e-2 on O(1) outputs, and
every simulated rank must hold bit-identical results."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(world, T, N, K, dtype, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    xs = [torch.randn((T, K), generator=g, device="cuda", dtype=torch.float32).to(dtype) for _ in range(world)]
    ws = [(torch.randn((N, K), generator=g, device="cuda", dtype=torch.float32) * K ** -0.5).to(dtype) for _ in range(world)]
    want = sum(torch.nn.functional.linear(x.float(), w.float()).to(dtype).float() for x, w in zip(xs, ws))
    return xs, ws, want


@pytest.mark.parametrize("world,T,N,K,dtype", [
    (2, 64, 4096, 2048, torch.bfloat16),      # Llama-3-8B o_proj at TP=2, decode batch 64
    (8, 64, 1024, 512, torch.bfloat16),       # TP=8 slice shape (N cut to keep all ranks' workgroups resident at once)
    (4, 1, 2048, 1024, torch.bfloat16),       # one token
    (4, 37, 512, 96, torch.bfloat16),         # ragged row block, K steps that do not divide over the waves
    (2, 200, 1024, 256, torch.float16),       # several row blocks, IEEE half
    (1, 16, 256, 64, torch.bfloat16),         # world 1: the reduction is the identity
])
def test_fused_oproj_matches_linear_plus_allreduce(world, T, N, K, dtype):
    from mi355_attn.tp import FusedRowParallel
    f = FusedRowParallel(N, t_cap=256, local_world=world, device="cuda:0")
    try:
        for call in range(3):                  # three calls: both buffer halves, growing epochs, a smaller batch last
            t = T if call < 2 else max(1, T // 2)
            xs, ws, want = _case(world, t, N, K, dtype, seed=call)
            xs = [torch.nn.functional.pad(x, (0, 8))[:, :K] for x in xs]          # row-strided inputs (views of a wider buffer)
            outs = f(xs, ws)
            torch.cuda.synchronize()
            f.check()
            for r in range(world):
                assert torch.equal(outs[r], outs[0]), f"rank {r} differs from rank 0 (call {call})"
            torch.testing.assert_close(outs[0].float(), want, atol=2e-2, rtol=2e-2)
    finally:
        f.close()


@pytest.mark.parametrize("device_epoch", [True, False], ids=["epoch_on_device", "epoch_from_host"])
def test_fused_oproj_both_epoch_forms(device_epoch):
    """The epoch a call carries comes from the host's count or from the device word that each launch advances itself:
    five calls each way (both buffer halves, several times) give the same results."""
    from mi355_attn.tp import FusedRowParallel
    world, T, N, K = 4, 48, 1024, 512
    f = FusedRowParallel(N, t_cap=64, local_world=world, device="cuda:0", device_epoch=device_epoch)
    try:
        for call in range(5):
            xs, ws, want = _case(world, T, N, K, torch.bfloat16, seed=10 + call)
            outs = f(xs, ws)
            torch.cuda.synchronize()
            f.check()
            assert all(torch.equal(o, outs[0]) for o in outs)
            torch.testing.assert_close(outs[0].float(), want, atol=2e-2, rtol=2e-2)
        if device_epoch:
            assert sorted(set(f.epoch_state.tolist())) == [0, 5]     # tiles of the 3 row blocks in use: five calls each
    finally:
        f.close()


def test_fused_oproj_replays_from_a_graph():
    """One captured launch, replayed with new inputs in the same buffers: the device-side epoch is what makes the
    arguments of every replay identical (bench_one_batch's captured decode step under TP uses it this way)."""
    from mi355_attn.tp import FusedRowParallel
    world, T, N, K = 2, 64, 2048, 1024
    f = FusedRowParallel(N, t_cap=64, local_world=world, device="cuda:0")
    try:
        xs, ws, want = _case(world, T, N, K, torch.bfloat16, seed=20)
        outs = [torch.empty((T, N), dtype=torch.bfloat16, device="cuda") for _ in range(world)]
        f(xs, ws, out=outs)                                  # eager once (epoch 1)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            f(xs, ws, out=outs)
        for rep in range(4):
            nx, _, _ = _case(world, T, N, K, torch.bfloat16, seed=30 + rep)
            for x, n in zip(xs, nx):
                x.copy_(n)
            want = sum(torch.nn.functional.linear(x.float(), w.float()).to(torch.bfloat16).float() for x, w in zip(xs, ws))
            g.replay()
            torch.cuda.synchronize()
            f.check()
            assert torch.equal(outs[0], outs[1])
            torch.testing.assert_close(outs[0].float(), want, atol=2e-2, rtol=2e-2)
        assert sorted(set(f.epoch_state.tolist())) == [5]
    finally:
        f.close()


def test_fused_oproj_refuses_bad_shapes():
    from mi355_attn import _lib
    from mi355_attn.tp import FusedRowParallel
    with pytest.raises(_lib.MI355AttnError):
        FusedRowParallel(100, local_world=2, device="cuda:0")                       # N % 64
    f = FusedRowParallel(128, t_cap=16, local_world=2, device="cuda:0")
    try:
        x = [torch.zeros((32, 64), dtype=torch.bfloat16, device="cuda") for _ in range(2)]
        w = [torch.zeros((128, 64), dtype=torch.bfloat16, device="cuda") for _ in range(2)]
        with pytest.raises(_lib.MI355AttnError, match="t_cap"):
            f(x, w)
        with pytest.raises(_lib.MI355AttnError):                                    # K % 32
            f([t[:16, :48] for t in x], [t[:, :48] for t in w])
    finally:
        f.close()


def test_a_peer_that_never_arrives_cannot_pass_for_a_result():
    """A world of two whose rank 1 is never launched: rank 0's bounded wait for rank 1's tile runs out (~1 s), the launch ENDS
    (no hang), the error word is set and the rows of the tile hold NaN - a give-up is impossible to miss
    (reference: dist.all_reduce would block, layers/linear.py:228-232).  The instance is not used again."""
    from mi355_attn import _lib
    from mi355_attn.tp import FusedRowParallel
    T, N, K = 16, 64, 64                                         # one tile: one workgroup spins, for about a second
    f = FusedRowParallel(N, t_cap=16, local_world=2, device_epoch=False)      # both ranks' buffers on this device
    xs, ws, _ = _case(2, T, N, K, torch.bfloat16, seed=11)
    out = torch.zeros((T, N), dtype=torch.bfloat16, device="cuda")
    arr = f._entries(2)
    arr[0].x, arr[0].x_row_stride, arr[0].w, arr[0].w_row_stride = xs[0].data_ptr(), xs[0].stride(0), ws[0].data_ptr(), ws[0].stride(0)
    arr[0].out, arr[0].out_row_stride = out.data_ptr(), out.stride(0)
    # n_ranks = 1 of world = 2: the multi-process form of the call, with only THIS rank's entry
    rc = f._L.m355_oproj_allreduce(arr, 1, 2, T, N, K, f.t_cap, 1, None, f.error.data_ptr(), _lib.dtype_code(torch.bfloat16),
                                   _lib.current_stream())
    assert rc == 0
    torch.cuda.synchronize()                                     # returns: the wait is bounded
    assert int(f.error.item()) == 1
    assert torch.isnan(out.float()).all()
    with pytest.raises(_lib.MI355AttnError, match="never arrived"):
        f.check()
    f.close()
