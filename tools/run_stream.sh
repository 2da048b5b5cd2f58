mkdir -p gpurun_out; LOG=gpurun_out/stream.log; : > $LOG
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 180 --timeout-method thread >> $LOG 2>&1 || { tail -40 $LOG; exit 1; }
for v in $1; do
  echo "--- stream variant $v" >> $LOG
  M355_STREAM_VARIANT=$v timeout -k 10 120 python -u bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['us_per_launch'], d['roofline']['achieved'], d['ms_per_step'])" >> $LOG 2>&1
done
tail -15 $LOG
