#!/bin/bash
# Dev library, both decode kernels forced (M355_DECODE_SHARE=1 share kernel, 0 range kernel + merge) on contiguous KV pools
# (--row-pad 0, the reference's layout) and on the padded ones: which kernel should serve a contiguous pool.
for a in "" "--heads 32,8,128" "--heads 16,4,128" "--heads 16,4,64" "--heads 8,2,128" "--len-scale 8" "--copies 8" "--heads 32,8,128 --len-scale 4" "--heads 64,8,128"; do
 for pad in "" "--row-pad 0"; do
  for sh in 1 0; do
   M355_LIB_PATH=dev M355_DECODE_SHARE=$sh timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras --no-pmc --steps 10 --warmup 3 $a $pad 2>/dev/null | A="$a $pad share=$sh" python3 -c "
import sys, json, os
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print(os.environ['A'], r['us_per_launch'], r['frac'])" || exit 1
  done
 done
done
