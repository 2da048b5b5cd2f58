#!/usr/bin/env python3
"""bench_one_batch - the reference's single-batch harness (minisglang/bench_one_batch.py) rebuilt around the
MI355X attention path, without a server.

Two modes, like SURVEY.md §8 a15 asks:

  --correctness-test   the reference's three-phase sequence (bench_one_batch.py:219-259): prefill the first
                       `cut_len` tokens of every prompt, then EXTEND with `prefix_ppns = page_table[i, :cut_len//ps]`
                       (:176-188; the cached prefix is re-used through the page table, as a RadixCache hit would),
                       then greedy decode.  With synthetic weights there is no text to print; instead the run is
                       checked against a single-shot prefill of the whole prompts (same logits within bf16 noise,
                       same greedy tokens) and the token ids are printed.
  latency (default)    for every batch_size x input_len x output_len (BenchArgs, :73-84): prefill latency, per-step
                       decode latency (median), decode throughput tok/s = B*(output_len-1) / sum(decode time), appended
                       as JSON lines to --result-filename.  The reference lacks this mode (its latency path was cut).

Weights are random (no checkpoints offline), `--model` picks the architecture shape.  TP: launch with
`python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 bench_one_batch.py ...` - one rank per
GPU, heads sharded, RCCL all-reduce after o_proj and down_proj.
"""
from __future__ import annotations

import argparse
import dataclasses
import json
import os
import statistics
import sys
import time
from typing import List, Tuple

import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

import mi355_attn as m  # noqa: E402
from harness.llama_shell import SHAPES, HarnessRunner  # noqa: E402


@dataclasses.dataclass
class BenchArgs:                                    # bench_one_batch.py:72-84
    run_name: str = "default"
    batch_size: Tuple[int, ...] = (1,)
    input_len: Tuple[int, ...] = (1024,)
    output_len: Tuple[int, ...] = (16,)
    result_filename: str = "result.jsonl"
    correctness_test: bool = False
    cut_len: int = 4
    model: str = "llama-3-8b"
    page_size: int = 16
    graph: bool = False                             # decode steps replayed from one captured HIP graph (GraphedDecode)
    fused_row_parallel: bool = False                # TP: o_proj / down_proj + all-reduce as one launch where it applies

    @staticmethod
    def add_cli_args(p: argparse.ArgumentParser):
        p.add_argument("--run-name", type=str, default=BenchArgs.run_name)
        p.add_argument("--batch-size", type=int, nargs="+", default=BenchArgs.batch_size)
        p.add_argument("--input-len", type=int, nargs="+", default=BenchArgs.input_len)
        p.add_argument("--output-len", type=int, nargs="+", default=BenchArgs.output_len)
        p.add_argument("--result-filename", type=str, default=BenchArgs.result_filename)
        p.add_argument("--correctness-test", action="store_true")
        p.add_argument("--cut-len", type=int, default=BenchArgs.cut_len)
        p.add_argument("--model", type=str, default=BenchArgs.model, choices=sorted(SHAPES))
        p.add_argument("--page-size", type=int, default=BenchArgs.page_size)
        p.add_argument("--graph", action="store_true", help="decode through one captured HIP graph per run")
        p.add_argument("--fused-row-parallel", action="store_true",
                       help="under TP: row-parallel GEMM + all-reduce in one launch (m355_oproj_allreduce) for decode-size inputs")


def synth_prompts(batch_size: int, input_len: int, vocab: int, seed: int = 0) -> List[List[int]]:
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, vocab, (batch_size, input_len), generator=g).tolist()


def make_reqs(prompts: List[List[int]], upto: int | None = None) -> List[m.Req]:
    reqs = []
    for i, ids in enumerate(prompts):
        r = m.Req(i, list(ids[:upto] if upto is not None else ids), m.SamplingParams(temperature=0, max_new_tokens=16))
        r.prefix_ppns = []
        r.fill_ids = list(r.origin_input_ids)       # bench_one_batch.py:169-170
        reqs.append(r)
    return reqs


@torch.no_grad()
def extend(reqs, runner):                           # bench_one_batch.py:194-204
    batch = m.Batch(reqs=reqs, page_manager=runner.page_manager, kvcache=runner.kvcache, device=runner.device)
    batch.prepare_for_prefill()
    logits, next_ids = runner.forward(batch)
    return next_ids, logits, batch


@torch.no_grad()
def decode(input_token_ids, batch, runner):         # bench_one_batch.py:207-214
    batch.output_ids = input_token_ids
    batch.prepare_for_decode()
    logits, next_ids = runner.forward(batch)
    return next_ids, logits


class GraphedDecode:
    """The decode loop with one step - metadata kernel, every layer, greedy sampling - captured ONCE as a HIP graph and
    replayed: a decode step at small batch is a few hundred launches of a few microseconds each, i.e. bound by the
    host's launch rate.  Possible because nothing in the path allocates, synchronises or needs host data per step:
    pages come from the device free list (m355_decode_prepare_ring), lengths / positions / slots / token ids live in
    fixed buffers that the step updates in place, and the plan is laid out once for the final token count.  The host
    only moves its mirrors (Batch.advance_host_for_decode).  Same kernels, same order: tokens identical to decode()."""

    def __init__(self, first_ids, batch, runner, n_steps: int):
        assert runner.kvcache.ring is not None, "graph mode needs KVCache(device_free_list=True)"
        # (under TP the step's all-reduces are captured with it: RCCL collectives record into a HIP graph like any kernel)
        self.batch, self.runner = batch, runner
        B = len(batch.reqs)
        dev = first_ids.device
        self.ids = first_ids.clone()
        self.seq_lens = batch.seq_lens.clone()
        self.pos, self.loc = torch.empty_like(self.seq_lens), torch.empty_like(self.seq_lens)
        self.out = torch.zeros((B, max(n_steps, 1)), dtype=torch.int64, device=dev)
        self.step = torch.zeros((1, 1), dtype=torch.int64, device=dev)
        batch.mode = m.Mode.DECODE
        batch.input_ids, batch.output_ids = self.ids, None
        batch.seq_lens, batch.positions, batch.out_cache_loc = self.seq_lens, self.pos, self.loc
        batch.cu_q = batch.row_of_token = None
        runner.attn_backend.decode_capacity_hint = sum(batch._lens_host()) + B * (n_steps + 1)
        self.taken = 0
        # step 1 eagerly on a side stream (allocator pools, workspaces, plan buffer, library handles) ...
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            batch.advance_host_for_decode()
            self._device_step()
        torch.cuda.current_stream().wait_stream(side)
        self.taken += 1
        # ... then step 2 is captured (nothing runs during capture) and run by its first replay
        self.graph = torch.cuda.CUDAGraph()
        if n_steps >= 2:
            batch.advance_host_for_decode()
            with torch.cuda.graph(self.graph):
                self._device_step()
            # the captured kernels hold raw pointers into the backend's workspace and plan buffer: no other forward may
            # replace them while this graph lives (a bigger prefill or batch then raises instead of freeing them)
            runner.attn_backend.buffers_pinned = True
            self.graph.replay()
            self.taken += 1

    @torch.no_grad()
    def _device_step(self):
        b, r = self.batch, self.runner
        m.ops.decode_prepare_ring(r.page_manager.page_table, b.page_table_ids, self.seq_lens, r.kvcache.ring, r.page_size,
                                  out=(self.seq_lens, self.pos, self.loc))
        _, nxt = r.forward(b)
        self.ids.copy_(nxt)
        self.out.scatter_(1, self.step.expand(self.out.shape[0], 1), nxt.unsqueeze(1))
        self.step += 1

    def next(self):
        """One more decode step (host mirrors + one graph launch).  Nothing in a replayed step looks at the fused
        epilogue's error word, so every 8 steps (every step at first) an asynchronous copy of it is started and the previous
        one examined."""
        self.batch.advance_host_for_decode()
        self.graph.replay()
        self.taken += 1
        fused = getattr(self.runner.model, "fused", None)
        if fused is not None and (self.taken % 8 == 0 or self.taken < 8):
            fused.poll_error()

    def tokens(self) -> torch.Tensor:
        return self.out[:, : self.taken]

    def release(self) -> None:
        """Drop the captured graph; the backend may size its buffers freely again."""
        torch.cuda.synchronize()
        self.runner.attn_backend.check()                 # (a sync was just paid: the decode plan's give-up word)
        self.graph = None
        self.runner.attn_backend.buffers_pinned = False
        self.runner.attn_backend.decode_capacity_hint = 0


def reset_pools(runner):
    """All pages and page-table rows back to their free lists (the harness owns every request)."""
    runner.kvcache.clear()
    runner.page_manager.clear()


REHEARSAL_NO_OVERLAP = 77       # exit code / stdout marker of a one-GPU rehearsal whose ranks were not run side by side


def _rehearsal_check(runner) -> None:
    """Two ranks rehearsing on ONE GPU (M355_BENCH_REHEARSE=1) with the fused row-parallel launches depend on the device
    running both processes' kernels side by side: a rank's launch waits, bounded, for its peer's tile.  When the device
    serves the processes one after the other instead, every such wait runs to its bound (seconds), the launches pile up on
    the stream and the run takes minutes before anything reports it.  So the rehearsal - and only the rehearsal: with a GPU
    per rank a wait only gives up when a peer is gone - looks at the give-up word after every decode step (a sync per step
    is fine in a correctness run) and ends at the first one, saying so."""
    fused = getattr(runner.model, "fused", None)
    if fused is None or os.environ.get("M355_BENCH_REHEARSE") != "1":
        return
    torch.cuda.synchronize()
    if int(fused.error.item()):
        print("REHEARSAL_NO_OVERLAP: a fused row-parallel wait gave up - this device did not run the two ranks' kernels side by side",
              flush=True)
        os._exit(REHEARSAL_NO_OVERLAP)


def three_phase(runner, prompts, cut_len: int, output_len: int, graph: bool = False):
    """prefill(cut) -> extend over the cached prefix -> decode; returns (final prefill logits, generated ids [B, output_len])."""
    ps = runner.page_size
    reqs = make_reqs(prompts, upto=cut_len)
    if cut_len > 0:
        extend(reqs, runner)
        # the reference leaks the first batch's page-table rows here; give them back so the same rows are re-used
        runner.page_manager.free([r.page_table_id for r in reqs])
    for i, r in enumerate(reqs):                    # bench_one_batch.py:176-188
        r.fill_ids = r.fill_ids + list(prompts[i][cut_len:])
        r.prefix_ppns = runner.page_manager.page_table[r.page_table_id, : cut_len // ps].tolist() if cut_len > 0 else []
    next_ids, logits, batch = extend(reqs, runner)
    if graph and output_len > 1:
        g = GraphedDecode(next_ids, batch, runner, output_len - 1)
        while g.taken < output_len - 1:
            g.next()
            _rehearsal_check(runner)
        toks = torch.cat([next_ids.unsqueeze(1), g.tokens()], dim=1)
        g.release()
        return logits, toks
    out = [next_ids]
    for _ in range(output_len - 1):
        next_ids, _ = decode(next_ids, batch, runner)
        out.append(next_ids)
        _rehearsal_check(runner)
    return logits, torch.stack(out, dim=1)


def correctness_test(runner, bench_args: BenchArgs, rank_print):
    shape = runner.model.shape
    B, L = bench_args.batch_size[0], bench_args.input_len[0]
    g = torch.Generator().manual_seed(1)
    lens = [int(x) for x in torch.randint(max(bench_args.cut_len + 1, L // 2), L + 1, (B,), generator=g)]
    prompts = [p[:n] for p, n in zip(synth_prompts(B, L, shape.vocab, seed=2), lens)]
    logits3, ids3 = three_phase(runner, prompts, bench_args.cut_len, bench_args.output_len[0])
    reset_pools(runner)
    logits1, ids1 = three_phase(runner, prompts, 0, bench_args.output_len[0])
    err = float((logits3.float() - logits1.float()).abs().max())
    ref = float(logits1.float().abs().max())
    same = float((ids3 == ids1).float().mean())
    rank_print(f"prompt lens {lens}, cut_len {bench_args.cut_len}")
    rank_print(f"prefill logits: cut+extend vs single shot max|diff| {err:.4f} (max |logit| {ref:.2f})")
    rank_print(f"greedy tokens agree: {100 * same:.1f}%")
    for i in range(min(B, 3)):
        rank_print(f"========== Prompt {i} ========== {ids3[i].tolist()}")
    if bench_args.graph:
        reset_pools(runner)
        _, idsg = three_phase(runner, prompts, bench_args.cut_len, bench_args.output_len[0], graph=True)
        gsame = bool(torch.equal(idsg, ids3))
        rank_print(f"graphed decode tokens identical to the eager loop: {gsame}")
        if not gsame:
            same = 0.0
    fused = getattr(runner.model, "fused", None)
    if fused is not None:
        # (two ranks rehearsing on ONE GPU depend on the device running both processes' kernels side by side; when it does not,
        # a bounded wait gives up after a second and the tokens that follow mean nothing)
        torch.cuda.synchronize()
        gave_up = fused.error.clone()
        if dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(gave_up, op=dist.ReduceOp.MAX)             # on ANY rank
        rank_print(f"fused row-parallel waits gave up: {bool(int(gave_up.item()))}")
    return err, ref, same


def latency_test(runner, bench_args: BenchArgs, bs: int, il: int, ol: int, rank_print, tp_size: int):
    shape = runner.model.shape
    sync = torch.cuda.synchronize
    result = None
    for it in range(2):                             # first pass warms up (allocator, hipBLASLt heuristics)
        reset_pools(runner)
        reqs = make_reqs(synth_prompts(bs, il, shape.vocab, seed=it))
        sync()
        t0 = time.perf_counter()
        next_ids, _, batch = extend(reqs, runner)
        sync()
        prefill = time.perf_counter() - t0
        steps = []
        if bench_args.graph and ol > 3:
            t0 = time.perf_counter()
            g = GraphedDecode(next_ids, batch, runner, ol - 1)       # runs steps 1 and 2 (eager warm-up, capture + replay)
            sync()
            setup = time.perf_counter() - t0
            while g.taken < ol - 1:
                t0 = time.perf_counter()
                g.next()
                sync()
                steps.append(time.perf_counter() - t0)
            g.release()
            total_decode = sum(steps) / len(steps) * (ol - 1)        # the two set-up steps priced as replayed steps
        else:
            setup = None
            for _ in range(ol - 1):
                t0 = time.perf_counter()
                next_ids, _ = decode(next_ids, batch, runner)
                sync()
                steps.append(time.perf_counter() - t0)
            total_decode = sum(steps)
        result = {
            "run_name": bench_args.run_name, "model": shape.name, "tp_size": tp_size, "page_size": runner.page_size,
            "batch_size": bs, "input_len": il, "output_len": ol, "weights": "synthetic", "dtype": "bf16",
            "prefill_latency": prefill, "prefill_throughput": bs * il / prefill,
            "median_decode_latency": statistics.median(steps) if steps else None,
            "median_decode_throughput": bs / statistics.median(steps) if steps else None,
            "total_decode_latency": total_decode,
            "decode_throughput": bs * (ol - 1) / total_decode if steps else None,
            "total_latency": prefill + total_decode, "overall_throughput": bs * (il + ol) / (prefill + total_decode),
            "decode_graph": bool(setup is not None), "graph_setup_latency": setup,
        }
    reset_pools(runner)
    rank_print(json.dumps(result))
    return result


def main():
    ap = argparse.ArgumentParser()
    BenchArgs.add_cli_args(ap)
    a = ap.parse_args()
    ba = BenchArgs(**{f.name: (tuple(getattr(a, f.name)) if isinstance(getattr(a, f.name), (list, tuple)) else getattr(a, f.name))
                      for f in dataclasses.fields(BenchArgs)})
    world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))
    rehearse = os.environ.get("M355_BENCH_REHEARSE") == "1"      # dev: every rank on cuda:0, gloo (a one-GPU box)
    if rehearse:
        local = 0
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    if world > 1:
        if rehearse:
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")      # one node: gloo must not go looking up the host name (it may not resolve)
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device(dev))
    rank_print = print if rank == 0 else (lambda *x, **k: None)
    shape = SHAPES[ba.model]
    max_bs, max_len = max(ba.batch_size), max(ba.input_len) + max(ba.output_len) + ba.page_size
    runner = HarnessRunner(shape, ba.page_size, max_total_tokens=max_bs * (max_len + ba.page_size) + 64 * ba.page_size,
                           max_reqs=max_bs, context_len=max_len, tp_rank=rank, tp_size=world, device=dev,
                           fused_row_parallel=ba.fused_row_parallel)
    if ba.correctness_test:
        err, ref, same = correctness_test(runner, ba, rank_print)
        ok = err <= 0.05 * max(ref, 1.0)
        rank_print("correctness:", "PASS" if ok else "FAIL")
        code = 0 if ok else 1
    else:
        code = 0
        for bs in ba.batch_size:
            for il in ba.input_len:
                for ol in ba.output_len:
                    res = latency_test(runner, ba, bs, il, ol, rank_print, world)
                    if rank == 0 and ba.result_filename:
                        with open(ba.result_filename, "a") as f:
                            f.write(json.dumps(res) + "\n")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    sys.exit(code)


if __name__ == "__main__":
    main()
