mkdir -p gpurun_out; LOG=gpurun_out/exp.log; : > $LOG
run() { echo "--- $*" >> $LOG; timeout -k 10 200 python -u bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['us_per_launch'], d['roofline']['achieved'], d['roofline']['algorithmic_bytes_per_launch'])" >> $LOG 2>&1; }
export M355_DECODE_VARIANT=1
run --chunk 256
run --heads 4,1,64 --copies 8 --chunk 256
run --heads 4,1,64 --copies 8 --chunk 512
run --heads 8,2,64 --copies 4 --chunk 256
run --heads 16,4,64 --copies 2 --chunk 256
export M355_DECODE_VARIANT=4
run --heads 4,1,64 --copies 8 --chunk 256
cat $LOG
