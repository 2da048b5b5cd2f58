#!/bin/bash
# The range-kernel (+ merge) workloads of DESIGN 3.3, product library, twice each: us per launch pair, fraction of 8 TB/s.
for a in "--heads 8,1,128" "--heads 8,2,128" "--heads 8,1,128 --fixed-len 32768 --batch 8" "--heads 8,1,128 --fixed-len 32768 --batch 1" "--fixed-len 2048 --batch 8" "--row-pad 0"; do
 for v in 0 1; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras --no-pmc --steps 10 --warmup 3 $a 2>/dev/null | A="$a" python3 -c "
import sys, json, os
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print(os.environ['A'], r['kernel'], r['us_per_launch'], r['frac'])" || exit 1
 done
done
