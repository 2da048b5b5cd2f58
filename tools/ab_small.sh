for a in "--heads 8,2,128" "--heads 8,1,128" "--heads 16,4,128" "--heads 8,1,128 --fixed-len 32768 --batch 8" "--heads 8,1,128 --fixed-len 32768 --batch 1" "--fixed-len 2048 --batch 8" "--heads 8,1,128 --fixed-len 32768 --batch 2" "--heads 8,1,128 --fixed-len 4096 --batch 16"; do
 for sh in 0 1; do
  M355_LIB_PATH=dev M355_DECODE_SHARE=$sh timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 $a 2>/dev/null | A="$a share=$sh" python3 -c "
import sys, json, os
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print(os.environ['A'], r['us_per_launch'], r['frac'])"
 done
done
