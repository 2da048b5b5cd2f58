#!/bin/bash
# The per-rank workloads of `bench.py --gpus 2/4/8` (weak scaling: heads / N, batch x N) on ONE GPU: product library, then the
# dev library with either decode kernel forced.  value = whole-step GB/s of the rank (N=1 headline for comparison first).
for a in "" "--heads 16,4,64 --copies 2" "--heads 8,2,64 --copies 4" "--heads 4,1,64 --copies 8"; do
 for mode in prod share range; do
  case $mode in prod) E="";; share) E="M355_LIB_PATH=dev M355_DECODE_SHARE=1";; range) E="M355_LIB_PATH=dev M355_DECODE_SHARE=0";; esac
  env $E timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras --no-pmc --steps 10 --warmup 3 $a 2>/dev/null | A="$a [$mode]" python3 -c "
import sys, json, os
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print(os.environ['A'], r['kernel'].split('::')[1][:14], 'us/launch', r['us_per_launch'], 'frac', r['frac'], 'step ms', d['ms_per_step'], 'value', d['value'])" || exit 1
 done
done
