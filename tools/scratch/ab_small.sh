run() { timeout -k 10 120 python -u bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('  %s  %.2f us  %.0f GB/s  frac %.3f  step %.3f ms' % (r['kernel'], r['us_per_launch'], r['achieved'], r['frac'], d['ms_per_step']))"; }
echo "cfg2"; run
echo "70B/TP8 shape (8,1,128), B=64"; run --heads 8,1,128
echo "8B/TP8 shape (4,1,128), B=64"; run --heads 4,1,128
echo "8B/TP4 (8,2,128), B=64"; run --heads 8,2,128
echo "8B/TP2 (16,4,128), B=64"; run --heads 16,4,128
echo "1B/TP8 (4,1,64), B=512"; run --heads 4,1,64 --copies 8
echo "1B/TP2 (16,4,64), B=128"; run --heads 16,4,64 --copies 2
echo "B=8 x 2048 1B"; run --fixed-len 2048 --batch 8
