#!/bin/bash
# The prefill workloads DESIGN 3.4 tabulates, one line each (us per layer, TFLOP/s), twice each.  usage (GPU box): tools/run_prefill_lines.sh
for a in "" "--heads 32,8,64" "--heads 8,1,128" "--suffix 256,256" "--groups 8 --per-group 4" "--prefix 4096 --suffix 64,256" \
         "--prefix 0 --suffix 2048,2048 --groups 1 --per-group 4 --layers 2 --reps 3" "--prefix 0 --suffix 8192,8192 --groups 1 --per-group 2 --layers 2 --reps 3"; do
 for v in 0 1; do
  echo -n "$a: "
  timeout -k 10 200 python tools/bench_prefill.py $a 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['us_per_layer'], d['tflops'])" || exit 1
 done
done
