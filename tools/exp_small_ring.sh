#!/bin/bash
# Dev library: ring depth and minimum range length of the range kernel on the small-launch shapes
for a in "--heads 8,1,128" "--heads 8,1,128 --fixed-len 32768 --batch 1" "--heads 8,1,128 --fixed-len 32768 --batch 8" "--heads 8,2,128" "--fixed-len 2048 --batch 8"; do
 for cfg in "2 128" "2 64" "2 96" "2 192" "2 256" "3 256"; do
  set -- $cfg
  M355_LIB_PATH=dev M355_MFMA_RING=$1 M355_MFMA_RMIN=$2 timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras --no-pmc --steps 10 --warmup 3 $a 2>/dev/null | A="$a ring=$1 rmin=$2" python3 -c "
import sys, json, os
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print(os.environ['A'], r['us_per_launch'], r['frac'])" || exit 1
 done
done
