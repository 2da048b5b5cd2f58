#!/bin/bash
# dev: roofline launch time of the planned decode, single-launch kernel (share=1) against per-wave ranges + merge (share=0)
run() { for sh in 0 1; do M355_DECODE_SHARE=$sh timeout -k 10 120 python -u bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('  share=$sh  %.2f us  %.0f GB/s  frac %.3f  step %.3f ms' % (r['us_per_launch'], r['achieved'], r['frac'], d['ms_per_step']))"; done; }
echo "cfg2 (1B shape, B=64)"; run
echo "8B shape (32,8,128), B=64"; run --heads 32,8,128
echo "cfg2 lengths x8"; run --len-scale 8
echo "cfg2 batch x8"; run --copies 8
echo "70B/TP8 shape (8,1,128), B=64"; run --heads 8,1,128
echo "70B/TP8 32k x B=32"; run --heads 8,1,128 --fixed-len 32768 --batch 32
echo "70B/TP8 32k x B=8"; run --heads 8,1,128 --fixed-len 32768 --batch 8
echo "TP4 of 8B (8,2,128), B=64"; run --heads 8,2,128
echo "B=8 x 2048, 1B shape"; run --fixed-len 2048 --batch 8
