#!/bin/bash
# Dev: prefill kernel geometry sweep (M355_PREFILL_WG = 10*waves + ring tiles/4)
set -e
for cfg in 41 42 82; do
  echo "== D=64 cfg $cfg"
  M355_PREFILL_WG=$cfg timeout -k 10 120 python tools/bench_prefill.py --heads 32,8,64
  M355_PREFILL_WG=$cfg timeout -k 10 120 python tools/bench_prefill.py --heads 32,8,64 --prefix 0 --suffix 8192,8192 --groups 1 --per-group 2 --layers 2 --reps 3
done
for cfg in 41 82; do
  echo "== D=128 cfg $cfg"
  M355_PREFILL_WG=$cfg timeout -k 10 120 python tools/bench_prefill.py
  M355_PREFILL_WG=$cfg timeout -k 10 120 python tools/bench_prefill.py --heads 32,8,128 --prefix 0 --suffix 8192,8192 --groups 1 --per-group 2 --layers 2 --reps 3
  M355_PREFILL_WG=$cfg timeout -k 10 120 python tools/bench_prefill.py --heads 8,1,128 --prefix 0 --suffix 32768,32768 --groups 1 --per-group 1 --layers 2 --reps 3
done
