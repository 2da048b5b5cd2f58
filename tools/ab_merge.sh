#!/bin/bash
# A/B on the dev library: pieces merged by their last arriver inside decode_mfma_kernel (M355_MERGE_IN_KERNEL=1) against the separate
# stream_merge_kernel launch (the product path).  usage (GPU box, repo root): tools/ab_merge.sh
for a in "--heads 8,1,128" "--heads 8,2,128" "--heads 16,4,128" "--heads 8,1,128 --fixed-len 32768 --batch 8" "--heads 8,1,128 --fixed-len 32768 --batch 1" \
         "--fixed-len 2048 --batch 8" "--heads 8,1,128 --fixed-len 32768 --batch 2" "--heads 8,1,128 --fixed-len 131072 --batch 1"; do
 for ml in 0 1; do
  M355_LIB_PATH=dev M355_MERGE_IN_KERNEL=$ml timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras --no-pmc --steps 10 --warmup 3 $a 2>/dev/null | A="$a merge_in_kernel=$ml" python3 -c "
import sys, json, os
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print(os.environ['A'], r['kernel'], r['us_per_launch'], r['frac'])" || exit 1
 done
done
