#!/usr/bin/env python3
"""bench.py - decode-attention hot path on MI355X (BASELINE.json configs[1]) + the rest of BASELINE's metric.

`python bench.py --gpus N --steps K --warmup W` prints ONE JSON line (rank 0).  With N > 1 and no WORLD_SIZE in the
environment it starts `python -m torch.distributed.run --nproc-per-node N bench.py ...` itself as a child process -
before anything touches the GPU - and relays the child's line (the reference spawns its TP ranks itself too,
bench_one_batch.py:275-292); launched by torch.distributed.run it is one rank per GPU over RCCL.

Workload of `value` at N=1: Llama-3.2-1B head shape (Hq=32, Hkv=8, D=64, 16 layers), batch 64 with ragged lengths
`random.seed(0); randint(64, 2048)` (sum 73 961), page_size 16, physical pages handed out in shuffled order,
distinct K/V pools per layer (2.4 GB: nothing is served from the 256 MiB Infinity Cache).

One STEP = one decode step of the hot path over the whole batch: Batch.prepare_for_decode (host plan + the
metadata kernel) and then, for each of the 16 layers, the KV slot write and paged decode attention (one fused
launch, m355_decode_store_attn_planned).  `value` = algorithmic HBM bytes of the attention path per step / step time,
summed over ranks (bytes_all = 4*Hkv*D*sum(L) + 4*B*Hq*D + 4*sum(ceil(L/ps)) + 16*B per layer, plus
the kv_store bytes).  Inputs are resident in HBM before the timed region.

N>1: the reference's tensor parallelism - every rank owns Hq/N query heads and Hkv/N kv heads of EVERY sequence
(models/llama.py:76-87) - with the global batch grown to 64*N so per-GPU bytes stay fixed ("weak").  The attention path
has no exchange step: the timed steps contain no collective.  The one collective of the surrounding
layer - the RCCL all-reduce of the o_proj output (layers/linear.py:228-232) - is timed separately and reported beside.

`roofline`: the decode launches alone, exactly as the steps make them, HIP-event timed over all layers back to back,
against 8.0 TB/s; the rocprofv3 average of the same command is profiles/rNN_bench_kernel_stats.csv.
`cpu_baseline`: the oracle (torch-native algorithm restated in oracle/attention_ref.py) on the host cores for one layer
of the same batch, rank 0, N=1 only.

`extras` (side measurements in the same run, each named; --no-extras skips them):
  N=1  decode_8b_shape, decode_cfg5_per_rank (70B/TP8 per-rank shape, 32k context, B = 1/8/32), prefill_cfg3,
       bench_one_batch_llama3_8b (eager and --graph), cpu_baseline_cfg1
  N>1  bench_one_batch_llama3_8b at TP=N, prefill_cfg3_per_rank, decode_cfg5_per_rank (N=8), the o_proj all-reduce legs
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import socket
import statistics
import subprocess
import sys
import time

import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.abspath(__file__))
for p in (REPO, os.path.join(REPO, "mini-sglang_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

MODEL = dict(name="Llama-3.2-1B", layers=16, hidden=2048, Hq=32, Hkv=8, D=64)
PAGE_SIZE = 16


def self_launch(args) -> int:
    """N > 1 without a launcher: become the launcher.  Nothing in this process has touched the GPU yet."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="only the headline measurement (profiling passes)")
    ap.add_argument("--no-pmc", action="store_true", help="do not measure roofline.traffic in this run (see measure_traffic())")
    ap.add_argument("--fused-probe", action="store_true", help="internal: child of an N>1 run, see fused_probe()")
    ap.add_argument("--one-batch-graph-child", action="store_true", help="internal: child of an N>1 run, see one_batch_graph_tp()")
    ap.add_argument("--heads", type=str, default="", help="experiment: Hq,Hkv,D per GPU instead of the 1B shape")
    ap.add_argument("--copies", type=int, default=0, help="experiment: batch = 64 * copies")
    ap.add_argument("--fixed-len", type=int, default=0, help="experiment: every request has this length (with --batch)")
    ap.add_argument("--batch", type=int, default=0, help="experiment: batch size for --fixed-len")
    ap.add_argument("--layers", type=int, default=0, help="experiment: number of layers (distinct KV pools)")
    ap.add_argument("--row-pad", type=int, default=-1, help="experiment: KV row padding in elements (0 = the reference's contiguous pools)")
    ap.add_argument("--len-scale", type=int, default=1, help="experiment: every length x this (SURVEY 8d cfg2 'long' variant: 8)")
    args = ap.parse_args()

    if args.fused_probe:
        sys.exit(fused_probe_child())
    if args.one_batch_graph_child:
        sys.exit(one_batch_graph_child())
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    rehearse = os.environ.get("M355_BENCH_REHEARSE") == "1"            # dev: all ranks on cuda:0 over gloo (one-GPU box)
    if rehearse:
        local_rank = 0
    experiment = bool(args.heads or args.copies or args.fixed_len or args.layers or args.len_scale > 1 or args.row_pad >= 0)
    measured = None
    if world == 1 and not experiment and not args.no_pmc:
        note(0, "HBM traffic: two rocprofv3 --pmc child passes of the headline leg")
        measured = measure_traffic()                   # child processes, BEFORE this process touches the GPU
        note(0, "HBM traffic measured" if measured else "HBM traffic not measured here: falling back to the committed summary")
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    if world > 1:
        if rehearse:
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")      # one node: gloo must not go looking up the host name (it may not resolve)
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device(dev))     # RCCL over xGMI

    from harness import workloads as W
    from mi355_attn.tp import shard_heads

    tp = world
    shard = shard_heads(MODEL["Hq"], MODEL["Hkv"], MODEL["D"], tp, rank)    # refuses tp > num_kv_heads
    Hq, Hkv, D, L = shard.num_heads, shard.num_kv_heads, MODEL["D"], MODEL["layers"]
    if args.heads:
        Hq, Hkv, D = (int(x) for x in args.heads.split(","))
    ps = PAGE_SIZE
    lens0 = W.cfg2_lengths(args.copies or tp)          # global batch 64 * tp
    if args.fixed_len:
        lens0 = [args.fixed_len] * (args.batch or 1)
    if args.len_scale > 1:
        lens0 = [l * args.len_scale for l in lens0]
    if args.layers:
        L = args.layers
    B = len(lens0)
    case = W.decode_case(dev, Hq, Hkv, D, lens0, L, ps=ps, room=args.warmup + args.steps + 16, seed=1234 + rank,
                         row_pad=args.row_pad if args.row_pad >= 0 else None, device_free_list=os.environ.get("M355_DEVICE_FREE_LIST", "1") != "0")
    batch, backend = case.batch, case.backend

    def fence():
        if tp > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        W.decode_step(case)
    fence()
    lens_start = list(batch.seq_lens_host)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        W.decode_step(case)
    fence()
    dt = time.perf_counter() - t0
    if tp > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    ms_per_step = dt / args.steps * 1e3
    step_bytes = 0                                     # bytes of step i: lengths are lens_start + i + 1
    for i in range(args.steps):
        cur = [l + i + 1 for l in lens_start]
        step_bytes += L * (W.attn_bytes_per_layer(cur, Hq, Hkv, D, ps) + W.kv_store_bytes_per_layer(B, Hkv, D))
    value = step_bytes * tp / dt / 1e9                 # every rank moves the same number of bytes

    # ---- roofline of the dominant kernel: the steps' decode launches only, all layers back to back
    lens_now = list(batch.seq_lens_host)
    us_per_launch, bytes_launch = W.time_decode_launches(case, reps=20)
    achieved = bytes_launch / us_per_launch / 1e3
    kernel = W.decode_kernel_name(case)
    traffic, traffic_source = (None, None) if (tp > 1 or experiment) else pmc_traffic(kernel, measured)

    cpu = None
    if rank == 0 and tp == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(case, lens_now)

    line = {
        "metric": "decode_attn_hbm_throughput", "value": round(value, 1), "unit": "GB/s", "n_gpus": tp,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": "Llama-3.2-1B decode attention path (kv_store + paged decode attn, 16 layers), "
                               f"batch {B} ragged 64..2048 (+ctx), page_size 16, shuffled pages",
                   "global_batch": B, "sum_seq_len": sum(lens_now), "page_size": ps, "layers": L,
                   "heads_per_gpu": [Hq, Hkv, D], "parallelism": f"tp{tp}", "decode_mode": backend.decode_mode},
        "pct_of_hbm_peak_whole_step": round(100 * value / tp / W.HBM_PEAK_GBPS, 2),
        "roofline": {"kernel": kernel, "bound": "hbm", "achieved": round(achieved, 1), "peak": W.HBM_PEAK_GBPS,
                     "unit": "GB/s", "frac": round(achieved / W.HBM_PEAK_GBPS, 4), "traffic": traffic,
                     "traffic_source": traffic_source, "us_per_launch": round(us_per_launch, 3),
                     "algorithmic_bytes_per_launch": bytes_launch, "launch": "attention + fused KV store"},
        "cpu_baseline": cpu,
    }
    del case, batch, backend
    torch.cuda.empty_cache()

    if not args.no_extras and not experiment:
        line["extras"] = extras(dev, tp, rank, rehearse, fence)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if tp > 1:
        dist.barrier()
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------
# side measurements: the rest of BASELINE.json's metric and configs, in the same driver-run record
# ------------------------------------------------------------------------------------------------
def guarded(fn, *a, **k):
    """A side measurement must not take the bench line down: its failure is reported in its own field.  Under TP every
    rank takes the same path through this (an exception on one rank only would leave the others in a collective)."""
    try:
        return fn(*a, **k)
    except Exception as e:      # noqa: BLE001
        return {"error": repr(e)[:300]}
    finally:
        torch.cuda.empty_cache()


def note(rank, msg):
    """Progress on stderr (rank 0): the run prints its one stdout line only at the end."""
    if rank == 0:
        print(f"[bench] {msg}", file=sys.stderr, flush=True)


def extras(dev, tp, rank, rehearse, fence) -> dict:
    from harness import workloads as W
    out = {}
    note(rank, "headline measured; side measurements: decode shapes")
    if tp == 1:
        # the 8B head shape on the same batch as the headline ((32, 8, 128))
        out["decode_8b_shape"] = guarded(W.decode_shape_report, dev, 32, 8, 128, W.cfg2_lengths(1), 16)
        # the headline batch on the REFERENCE's contiguous [S, Hkv, D] pools (memory/kvcache.py:24-35; what INTEGRATION.md's
        # one-line edit runs on): mi355_attn.KVCache pads every token row by 128 bytes (12.5 % of the pool on this shape)
        out["decode_contiguous_pool"] = guarded(W.decode_shape_report, dev, MODEL["Hq"], MODEL["Hkv"], MODEL["D"], W.cfg2_lengths(1), 16,
                                                row_pad=0)
        if isinstance(out["decode_contiguous_pool"], dict):
            out["decode_contiguous_pool"]["note"] = ("same batch as the headline on KVCache(row_pad=0), the reference's pool layout: the library serves "
                                                     "head_dim-64 batches on contiguous power-of-two rows with the range kernel + merge (the share kernel "
                                                     "takes 34.5 us there); the headline's padded rows cost 12.5 % more HBM per cached token on this shape")
    if tp in (1, 8):
        # cfg5: what ONE rank of Llama-3-70B at TP=8 streams - (8, 1, 128), every request 32 768 tokens (16 of the 80 layers)
        out["decode_cfg5_per_rank"] = {
            f"B{b}": guarded(W.decode_shape_report, dev, 8, 1, 128, [32768] * b, 16, reps=10) for b in (1, 8, 32)}
        out["decode_cfg5_per_rank"]["note"] = ("per-rank shape of Llama-3-70B at TP=8, context 32 768, page_size 16, "
                                               + ("one rank of 8 measured" if tp == 8 else "measured on one GPU"))
    # cfg3 / cfg4: Llama-3-8B prefill over shared cached prefixes, this rank's heads
    note(rank, "prefill cfg3")
    out["prefill_cfg3" if tp == 1 else "prefill_cfg3_per_rank"] = guarded(prefill_cfg3, dev, tp)
    if tp == 1:
        note(rank, "cfg3 end to end through the radix tree")
        guarded(W.cfg3_flow, dev, True)                      # untimed pass: first-use costs of this batch shape (module loads, pinned buffers)
        out["cfg3_flow"] = {"native_tree": guarded(W.cfg3_flow, dev, True), "python_tree": guarded(W.cfg3_flow, dev, False),
                            "protocol": "one untimed pass of the whole flow first; host times are wall clock, GPU times HIP events"}
    note(rank, "fused o_proj probe (child processes)" if tp > 1 else "bench_one_batch")
    fused_ok = tp > 1 and fused_probe(dev, tp)
    if tp > 1:
        out["o_proj_fused_probe"] = "ok" if fused_ok else "failed in its child processes: the fused legs below were skipped"
    if tp > 1:
        note(rank, f"probe {'ok' if fused_ok else 'FAILED'}; bench_one_batch")
    out["bench_one_batch_llama3_8b"] = guarded(one_batch, dev, tp, rank, rehearse, fused_ok)
    if tp > 1:
        note(rank, "bench_one_batch with the decode step captured as a graph (child processes)")
        if isinstance(out["bench_one_batch_llama3_8b"], dict):
            out["bench_one_batch_llama3_8b"].update(one_batch_graph_tp(dev, tp, rank, fused_ok))
    note(rank, "bench_one_batch done")
    if tp > 1:
        out.update(guarded(collective_legs, dev, tp, rank, rehearse, fence, fused_ok))
    if tp == 1:
        out["o_proj_fused_single_gpu"] = guarded(o_proj_single_gpu, dev)
    if tp == 1 and rank == 0:
        out["cpu_baseline_cfg1"] = guarded(cpu_baseline_cfg1)
    return out


def fused_probe(dev, tp) -> bool:
    """The one-launch o_proj + reduce writes through peer pointers, and a bad peer mapping is a GPU memory fault, which no
    `except` catches.  So its first run across these GPUs happens in CHILD processes (one per rank, their own gloo group
    on the next port, same devices): if a child faults, gives up waiting or disagrees with the plain sum, the ranks of
    this run skip the fused legs and the bench line survives."""
    env = child_env(23)
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--fused-probe"], env=env, capture_output=True, text=True,
                           timeout=120)
        ok = r.returncode == 0 and "FUSED_PROBE_OK" in r.stdout
    except Exception:           # noqa: BLE001  (timeout, spawn failure)
        ok = False
    flag = torch.tensor([1 if ok else 0], device=dev, dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return bool(int(flag.item()))


def fused_probe_child() -> int:
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local_rank = 0 if os.environ.get("M355_BENCH_REHEARSE") == "1" else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    import datetime
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")      # one node: gloo must not go looking up the host name (it may not resolve)
    dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=90))
    from mi355_attn.tp import FusedRowParallel
    hidden, T = 4096, 64
    g = torch.Generator(device=dev).manual_seed(77 + rank)
    x = torch.randn((T, hidden // world), generator=g, device=dev).to(torch.bfloat16)
    w = (torch.randn((hidden, hidden // world), generator=g, device=dev) * 0.02).to(torch.bfloat16)
    f = FusedRowParallel(hidden, t_cap=256, device=dev)
    try:
        for _ in range(3):                                  # epochs, both buffer halves
            y = f(x, w)
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()                            # an exchange is ~20 us when the ranks' kernels run side by side; a
        for _ in range(4):                                  # device that time-slices the processes shows its quantum here
            y = f(x, w)
        torch.cuda.synchronize()
        per_call_ms = (time.perf_counter() - t0) * 1e3 / 4
        want = torch.nn.functional.linear(x, w).float().cpu()
        dist.all_reduce(want)
        err = float((y.float().cpu() - want).abs().max())
        gave_up = int(f.error.item())
        dist.barrier()
    finally:
        f.close()
    dist.destroy_process_group()
    if gave_up or not err < 0.05:
        print(f"FUSED_PROBE_FAILED rank {rank}: max diff {err}, wait gave up {gave_up}", flush=True)
        return 1
    print(f"FUSED_PROBE_MS rank {rank}: {per_call_ms:.3f}", flush=True)
    print("FUSED_PROBE_OK", flush=True)
    return 0


def o_proj_single_gpu(dev):
    """The GEMM + push + reduce launch of SURVEY §8 f3 with a world of one (no peer: its own slot only), against
    F.linear, at the per-rank o_proj shapes of Llama-3-8B under TP 8 / 2 / 1, decode batch 64."""
    from mi355_attn.tp import FusedRowParallel
    res = {"what": "m355_oproj_allreduce, world 1, x [64, K] @ w [4096, K]^T bf16 vs torch matmul (hipBLASLt): GPU us per call, 20 calls replayed from a HIP graph"}
    f = FusedRowParallel(4096, t_cap=256, local_world=1, device=dev)
    try:
        for K in (512, 2048, 4096):
            x = torch.randn((64, K), device=dev, dtype=torch.bfloat16)
            w = torch.randn((4096, K), device=dev, dtype=torch.bfloat16) * 0.02
            t = {}
            o = [torch.empty((64, 4096), device=dev, dtype=torch.bfloat16)]
            for name, fn in (("fused", lambda: f([x], [w], out=o)), ("linear", lambda: torch.nn.functional.linear(x, w, out=o[0]) if False else torch.matmul(x, w.t(), out=o[0]))):
                fn()
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()                # 20 calls replayed from a graph: GPU time, not Python's launch rate
                with torch.cuda.graph(g):
                    for _ in range(20):
                        fn()
                g.replay()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    g.replay()
                e1.record()
                torch.cuda.synchronize()
                t[name] = round(e0.elapsed_time(e1) * 1e3 / 100, 2)
            res[f"K{K}"] = t
    finally:
        f.close()
    return res


def prefill_cfg3(dev, tp):
    from harness import workloads as W
    c = W.prefill_cfg3_case(dev, Hq=32 // tp, Hkv=8 // tp, D=128, layers=8)
    r = W.time_prefill(c, reps=10)
    r["workload"] = "Llama-3-8B shape, 4 groups x 4 requests, 1024-token shared cached prefix + 128..512 new tokens"
    return r


def one_batch(dev, tp, rank, rehearse, fused_ok=False):
    """bench_one_batch's latency mode (reference bench_one_batch.py:72-84, 195-214) on Llama-3-8B with synthetic weights:
    B=64, input 1024, 32 output tokens; tok/s = B * decode steps / sum(decode step time).  TP=N under torch.distributed."""
    import bench_one_batch as ob
    from harness.llama_shell import SHAPES, HarnessRunner
    c = ONE_BATCH_REHEARSAL if rehearse else ONE_BATCH_TP  # (a gloo rehearsal cannot all-reduce 8B-size activations in minutes)
    model, bs, il, ol = c["model"], c["bs"], c["il"], c["ol"]
    shape = SHAPES[model]
    runner = HarnessRunner(shape, PAGE_SIZE, max_total_tokens=bs * (il + ol + 2 * PAGE_SIZE) + 64 * PAGE_SIZE, max_reqs=bs,
                           context_len=il + ol + PAGE_SIZE, tp_rank=rank, tp_size=tp, device=dev, fused_row_parallel=fused_ok)
    quiet = lambda *a, **k: None      # noqa: E731
    res = {"model": shape.name, "weights": "synthetic", "dtype": "bf16", "tp_size": tp, "batch_size": bs, "input_len": il,
           "output_len": ol, "page_size": PAGE_SIZE}
    fused = runner.model.fused
    # TP=1: the eager loop and the captured-graph loop; TP=N: RCCL all-reduce after o_proj / down_proj, and the same run
    # with the two as fused GEMM + one-shot reduce launches for the decode steps
    for name, graph, use_fused in ((("eager", False, False), ("graph", True, False)) if tp == 1 else
                                   (("eager_rccl", False, False),) + ((("eager_fused_oproj", False, True),) if fused_ok else ())):
        runner.model.fused = fused if use_fused else None
        r = ob.latency_test(runner, ob.BenchArgs(graph=graph, model=model), bs, il, ol, quiet, tp)
        res[name] = {"prefill_tok_per_s": round(r["prefill_throughput"], 1), "prefill_latency_ms": round(r["prefill_latency"] * 1e3, 2),
                     "decode_tok_per_s": round(r["decode_throughput"], 1),
                     "median_decode_step_ms": round(r["median_decode_latency"] * 1e3, 3)}
    if fused is not None:
        torch.cuda.synchronize()
        res["fused_oproj_wait_gave_up"] = bool(int(fused.error.item()))
        fused.close()
    return res


ONE_BATCH_TP = dict(model="llama-3-8b", bs=64, il=1024, ol=32)
ONE_BATCH_REHEARSAL = dict(model="tiny", bs=4, il=64, ol=8)


def child_env(port_offset: int) -> dict:
    """Environment of a child rank: this rank's, on its own rendezvous port, without the launcher's store settings."""
    env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_")}
    env["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29500")) + port_offset)
    return env


def one_batch_graph_tp(dev, tp, rank, fused_ok):
    """bench_one_batch at TP = N with the decode step - every layer, its all-reduces, sampling - replayed from a HIP graph
    (eager decode at TP = 8 is bound by the host's launch rate, not by the GPUs): `graph_rccl` with RCCL all-reduces captured,
    `graph_fused_oproj` with the one-launch GEMM + reduce where it applies (T <= 64, K <= 2048 per rank).  Capturing
    collectives is the one thing here a one-GPU box cannot rehearse, so each leg runs in child processes (one per rank, own
    process group): a capture that raises, hangs or dies costs that field, not the bench line."""
    out = {}
    for name, fused, port in (("graph_rccl", False, 29), ("graph_fused_oproj", True, 31)):
        if fused and not fused_ok:
            continue
        note(rank, f"  {name} (children, at most 170 s)")
        try:
            env = child_env(port)
            env["M355_CHILD_FUSED"] = "1" if fused else "0"
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one-batch-graph-child"], env=env,
                               capture_output=True, text=True, timeout=170)
            lines = [l for l in r.stdout.splitlines() if l.startswith("ONE_BATCH_CHILD ")]
            res = json.loads(lines[-1][len("ONE_BATCH_CHILD "):]) if (r.returncode == 0 and lines) else None
            err = None if r.returncode == 0 else (r.stderr.strip().splitlines() or ["?"])[-1][:300]
        except Exception as e:      # noqa: BLE001  (timeout, spawn failure)
            res, err = None, repr(e)[:300]
        ok = torch.tensor([1 if err is None else 0], device=dev, dtype=torch.int32)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if not int(ok.item()):
            out[name] = {"error": "a child rank failed" + (f": {err}" if err else "")}
        else:
            out[name] = res if res is not None else {"note": "measured by rank 0's child"}
    return out


def one_batch_graph_child() -> int:
    import datetime
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    rehearse = os.environ.get("M355_BENCH_REHEARSE") == "1"
    fused = os.environ.get("M355_CHILD_FUSED") == "1"
    local_rank = 0 if rehearse else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    if rehearse:
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")      # one node: gloo must not go looking up the host name (it may not resolve)
        dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=120))
    else:
        dist.init_process_group("nccl", device_id=torch.device(dev), timeout=datetime.timedelta(seconds=150))
    import bench_one_batch as ob
    from harness.llama_shell import SHAPES, HarnessRunner
    c = ONE_BATCH_REHEARSAL if rehearse else ONE_BATCH_TP
    shape = SHAPES[c["model"]]
    bs, il, ol = c["bs"], c["il"], c["ol"]
    runner = HarnessRunner(shape, PAGE_SIZE, max_total_tokens=bs * (il + ol + 2 * PAGE_SIZE) + 64 * PAGE_SIZE, max_reqs=bs,
                           context_len=il + ol + PAGE_SIZE, tp_rank=rank, tp_size=world, device=dev, fused_row_parallel=fused)
    # (a gloo all-reduce cannot be captured: the rehearsal runs the RCCL leg as an eager loop, plumbing only; its tiny
    # model's row-parallel layers all take the fused launch, so the fused leg IS captured there)
    graph = fused or not rehearse
    r = ob.latency_test(runner, ob.BenchArgs(graph=graph, model=c["model"]), bs, il, ol, lambda *a, **k: None, world)
    torch.cuda.synchronize()
    gave_up = bool(int(runner.model.fused.error.item())) if runner.model.fused is not None else False
    dist.barrier()
    if rank == 0:
        print("ONE_BATCH_CHILD " + json.dumps({
            "decode_graph": bool(r["decode_graph"]), "prefill_tok_per_s": round(r["prefill_throughput"], 1),
            "prefill_latency_ms": round(r["prefill_latency"] * 1e3, 2), "decode_tok_per_s": round(r["decode_throughput"], 1),
            "median_decode_step_ms": round(r["median_decode_latency"] * 1e3, 3), "fused_oproj_wait_gave_up": gave_up}), flush=True)
    if runner.model.fused is not None:
        runner.model.fused.close()
    dist.destroy_process_group()
    return 0 if not gave_up else 1


def collective_legs(dev, tp, rank, rehearse, fence, fused_ok=False):
    """The layer's one exchange step, outside the path (layers/linear.py:228-232): RCCL all-reduce of the o_proj output at
    the decode and prefill sizes of Llama-3-8B, and o_proj GEMM + all-reduce back to back against the
    f3 variants (mi355_attn.tp)."""
    from mi355_attn.tp import all_reduce_rows, row_parallel_linear

    def timed(fn, reps):
        for _ in range(3):
            fn()
        fence()
        a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a0.record()
        for _ in range(reps):
            fn()
        a1.record()
        torch.cuda.synchronize()
        us = torch.tensor([a0.elapsed_time(a1) * 1e3 / reps], device=dev, dtype=torch.float64)
        dist.all_reduce(us, op=dist.ReduceOp.MAX)
        return float(us.item())

    def allreduce(T, hidden, reps):
        t = torch.zeros((T, hidden), dtype=torch.bfloat16, device=dev)
        us = timed(lambda: all_reduce_rows(t), reps)
        nbytes = t.numel() * 2
        return {"tensor": f"[{T}, {hidden}] bf16", "bytes": nbytes, "us": round(us, 2), "algbw_GBps": round(nbytes / us / 1e3, 1),
                "busbw_GBps": round(nbytes / us / 1e3 * 2 * (tp - 1) / tp, 1)}

    def o_proj(T, chunks, reps):
        k_in = 4096 // tp
        x = torch.randn((T, k_in), device=dev, dtype=torch.bfloat16)
        w = torch.randn((4096, k_in), device=dev, dtype=torch.bfloat16) * 0.02
        return round(timed(lambda: row_parallel_linear(x, w, chunks), reps), 2)

    def fused_o_proj(T, hidden, reps):
        from mi355_attn.tp import FusedRowParallel
        if not fused_ok:
            return {"skipped": "o_proj_fused_probe failed"}
        k_in = hidden // tp
        x = torch.randn((T, k_in), device=dev, dtype=torch.bfloat16)
        w = torch.randn((hidden, k_in), device=dev, dtype=torch.bfloat16) * 0.02
        f = FusedRowParallel(hidden, t_cap=256, device=dev)
        try:
            us = timed(lambda: f(x, w), reps)
            y = f(x, w)
            want = all_reduce_rows(torch.nn.functional.linear(x, w))
            err = float((y.float() - want.float()).abs().max())
            return {"us": round(us, 2), "max_abs_diff_vs_linear_plus_rccl": round(err, 4), "wait_gave_up": bool(int(f.error.item()))}
        finally:
            f.close()

    if rehearse:
        return {"allreduce_o_proj_llama3_8b": {"decode_B64": allreduce(64, 256, 2)},
                "o_proj_fused_oneshot": {"T8": guarded(fused_o_proj, 8, 256, 2)},
                "o_proj_all_reduce": {"us": {"T64": {"chunks1": o_proj(64, 1, 2), "chunks2": o_proj(64, 2, 2)}}}}
    return {
        "allreduce_o_proj_llama3_8b": {"decode_B64": allreduce(64, 4096, 50), "prefill_T4863": allreduce(4863, 4096, 10),
                                       "link_peak_GBps": "7 xGMI links x ~153 GB/s per GPU"},
        "o_proj_fused_oneshot": {"what": "GEMM + push to all ranks over xGMI + rank-order reduce in ONE launch (m355_oproj_allreduce) "
                                         "against o_proj_all_reduce.us.T64.chunks1 (F.linear + RCCL)",
                                 "T64": guarded(fused_o_proj, 64, 4096, 50), "T16": guarded(fused_o_proj, 16, 4096, 50)},
        "o_proj_all_reduce": {"shape": f"x [T, {4096 // tp}] @ w [4096, {4096 // tp}]^T bf16, SUM over {tp} ranks",
                              "us": {"T64": {f"chunks{c}": o_proj(64, c, 50) for c in (1, 2)},
                                     "T4863": {f"chunks{c}": o_proj(4863, c, 10) for c in (1, 2, 4, 8)}}}}


PMC_CHILD = ["--no-cpu-baseline", "--no-extras", "--no-pmc", "--steps", "4", "--warmup", "2"]


def measure_traffic():
    """HBM bytes per launch of this command's kernels, measured by THIS run: the headline leg is run twice more as a child
    under `rocprofv3 --pmc FETCH_SIZE` and `rocprofv3 --pmc WRITE_SIZE` (a pass of its own each, nothing but --pmc, as
    MI355X_MICROARCH.md's HBM section prescribes), before this process touches the GPU.  Both counters are in KiB; on gfx950
    FETCH_SIZE tallies 128-byte requests at 64 bytes, so reads are doubled.  Returns {kernel name: bytes per launch} plus a
    "source" entry, or None (no rocprofv3, already running under a profiler, a pass failed): pmc_traffic() then falls back to
    the newest committed summary and says so."""
    import shutil
    import sqlite3
    import tempfile
    if shutil.which("rocprofv3") is None or shutil.which("timeout") is None:
        return None
    if any(k.startswith("ROCPROF") for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return None                                    # this run is itself being profiled: no nesting
    acc = {}
    with tempfile.TemporaryDirectory(prefix="m355pmc") as tmp:
        env = dict(os.environ, TMPDIR=tmp)
        for leg, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
            out = os.path.join(tmp, leg)
            cmd = ["timeout", "-s", "KILL", "150", "rocprofv3", "--pmc", counter, "-d", out, "-o", leg, "--",
                   sys.executable, os.path.abspath(__file__)] + PMC_CHILD
            try:
                r = subprocess.run(cmd, env=env, cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=170)
            except Exception:                           # noqa: BLE001
                return None
            dbs = glob.glob(os.path.join(out, "**", "*_results.db"), recursive=True)
            if r.returncode != 0 or not dbs:
                return None
            try:
                db = sqlite3.connect(dbs[0])
                rows = db.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name = ? "
                                  "and kernel_name like '%m355::%' group by kernel_name", (counter,)).fetchall()
                db.close()
            except Exception:                           # noqa: BLE001
                return None
            if not rows:
                return None
            for name, _calls, kib in rows:
                acc[name] = acc.get(name, 0) + int(kib * 1024 * (2 if counter == "FETCH_SIZE" else 1))
    acc["source"] = ("measured by this run: child passes `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (one pass each) of "
                     "`bench.py " + " ".join(PMC_CHILD) + "`; KiB*1024, reads doubled (gfx950)")
    return acc


def pmc_traffic(kernel: str, measured=None):
    """HBM bytes per attention launch: from this run's own PMC passes (measure_traffic) when they ran, else from the newest
    committed summary of the same command (profiles/rNN_pmc_traffic.json, tools/summarize_profile.py).  The second value says
    which."""
    want = [w.split("::")[-1].strip(" ()+") for w in kernel.split("(+") if w.strip()]
    if measured:
        tot, ok = 0, True
        for w in want:
            hit = [v for k, v in measured.items() if k != "source" and w in k]
            ok = ok and bool(hit)
            tot += hit[0] if hit else 0
        if ok:
            return tot, measured["source"]
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc_traffic.json")))
    for path in reversed(files):
        with open(path) as f:
            kernels = json.load(f)["kernels"]
        tot, ok = 0, True
        for w in want:
            hit = [e["hbm_bytes_per_launch"] for k, e in kernels.items() if w in k]
            ok = ok and bool(hit)
            tot += hit[0] if hit else 0
        if ok:
            return tot, "profiles/" + os.path.basename(path) + " (separate rocprofv3 --pmc passes of this command, not this run)"
    return None, None


def cpu_baseline(case, lens):
    """The oracle's decode attention (per-sequence gather + SDPA, reference torch_native_backend.py:121-193) on the
    host cores: ONE layer of the same batch, 3 warm-ups + median of 10."""
    from harness import workloads as W
    from oracle import attention_ref as A
    n_thr = torch.get_num_threads()
    kp, vp = (t.cpu().contiguous() for t in case.kv.get_kv_cache(0))
    table = case.pm.page_table.cpu()
    rows, sl = case.batch.page_table_ids.cpu(), case.batch.seq_lens.cpu()
    qc = case.q.cpu().reshape(len(lens), case.Hq, case.D)
    for _ in range(3):
        A.decode_attn_ref(qc, kp, vp, table, rows, sl, case.ps)
    ts = []
    for _ in range(10):
        t0 = time.perf_counter()
        A.decode_attn_ref(qc, kp, vp, table, rows, sl, case.ps)
        ts.append(time.perf_counter() - t0)
    med = statistics.median(ts)
    return {"value": round(W.attn_bytes_per_layer(lens, case.Hq, case.Hkv, case.D, case.ps) / med / 1e9, 3), "unit": "GB/s",
            "cores": n_thr, "kind": "port",
            "sample": f"decode attention of 1 of 16 layers, same batch, bf16, 3 warm-ups + median of 10 ({med * 1e3:.1f} ms/layer)"}


def cpu_baseline_cfg1():
    """BASELINE.json configs[0] on the host cores, through the oracle (the reference's CPU-runnable case): Llama-3.2-1B
    head shape, page_size 1, batch 1, 128-token prompt (`arange(128)`), then 16 decode steps; metadata + attention per
    layer, 3 warm-ups + median of 10.  Weights play no part in the path, so tok/s here is the attention
    path alone over the model's 16 layers."""
    import numpy as np
    from oracle import attention_ref as A
    from oracle import memory_ref as M
    Hq, Hkv, D, L, ps, P, steps = 32, 8, 64, 16, 1, 128, 16
    g = torch.Generator().manual_seed(0)
    S = 512
    kp = torch.randn((S, Hkv, D), generator=g).to(torch.bfloat16)
    vp = torch.randn((S, Hkv, D), generator=g).to(torch.bfloat16)
    qkv = torch.randn((P + steps, (Hq + 2 * Hkv) * D), generator=g).to(torch.bfloat16)

    def med(fn, warm=3, n=10):
        for _ in range(warm):
            fn()
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return statistics.median(ts)

    def fresh():
        return M.RefPagePool(S, ps), M.RefPageTable(ps, 8, 131076)

    def prefill_meta():
        pool, pt = fresh()
        return M.prepare_for_prefill_ref([M.RefReq(list(range(P)))], pool, pt), pool, pt

    t_prep_prefill = med(lambda: prefill_meta())
    meta, pool, pt = prefill_meta()
    table = torch.from_numpy(pt.table)
    rows, sl, pl = (torch.from_numpy(np.asarray(x)) for x in (meta.page_table_ids, meta.seq_lens, meta.prefix_lens))
    q, k, v = (t.reshape(P + steps, -1, D) for t in qkv.split([Hq * D, Hkv * D, Hkv * D], dim=-1))
    loc = torch.from_numpy(meta.out_cache_loc)

    def extend_layer():
        A.write_kv_ref(kp, vp, loc, k[:P], v[:P])
        return A.extend_attn_ref(q[:P], kp, vp, table, rows, sl, pl, ps)

    t_extend = med(extend_layer)
    t_prep_decode, t_decode = [], []
    for s in range(steps):
        t0 = time.perf_counter()
        meta = M.prepare_for_decode_ref(meta, [0], pool, pt)
        t_prep_decode.append(time.perf_counter() - t0)
        table = torch.from_numpy(pt.table)
        sl = torch.from_numpy(np.asarray(meta.seq_lens))
        dloc = torch.from_numpy(meta.out_cache_loc)

        def decode_layer():
            A.write_kv_ref(kp, vp, dloc, k[P + s:P + s + 1], v[P + s:P + s + 1])
            return A.decode_attn_ref(q[P + s:P + s + 1], kp, vp, table, rows, sl, ps)

        t_decode.append(med(decode_layer, warm=1, n=5) if s else med(decode_layer))
    dec_layer = statistics.median(t_decode)
    step_s = statistics.median(t_prep_decode) + L * dec_layer
    return {"workload": "Llama-3.2-1B head shape, CPU, page_size 1, batch 1, 128-token prompt + 16 decode steps (oracle)",
            "cores": torch.get_num_threads(), "kind": "port",
            "prepare_for_prefill_ms": round(t_prep_prefill * 1e3, 3), "extend_attn_ms_per_layer": round(t_extend * 1e3, 3),
            "prepare_for_decode_ms": round(statistics.median(t_prep_decode) * 1e3, 3),
            "decode_attn_ms_per_layer": round(dec_layer * 1e3, 4),
            "decode_attn_path_tok_per_s": round(1.0 / step_s, 1),
            "decode_kv_GBps": round(4 * Hkv * D * (P + steps // 2) / dec_layer / 1e9, 3),
            "protocol": "3 warm-ups + median of 10 (decode: median over the 16 steps of per-step medians)"}


if __name__ == "__main__":
    main()
