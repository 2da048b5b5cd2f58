#!/bin/bash
# The share-kernel workloads of DESIGN 3.3, product library, twice each: us per launch, fraction of 8 TB/s.
for a in "" "--heads 32,8,128" "--len-scale 8" "--copies 8" "--heads 8,1,128 --fixed-len 32768 --batch 32" "--heads 16,4,128"; do
 for v in 0 1; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras --no-pmc --steps 10 --warmup 3 $a 2>/dev/null | A="$a" python3 -c "
import sys, json, os
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print(os.environ['A'], r['kernel'], r['us_per_launch'], r['frac'])" || exit 1
 done
done
