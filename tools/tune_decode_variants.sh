# Dev tool: correctness + timing sweep over the decode kernel variants (M355_DECODE_VARIANT), on the GPU box.
# usage: bash tools/tune_decode_variants.sh "<variants>" "<chunks>" [rocprof_variants]
mkdir -p gpurun_out
LOG=gpurun_out/tune.log
: > $LOG
for v in $1; do
  echo "=== variant $v tests" >> $LOG
  M355_DECODE_VARIANT=$v timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 120 --timeout-method thread -k "decode or flow or prefill" >> $LOG 2>&1 || { echo "TESTS FAILED variant $v" >> $LOG; continue; }
  for c in $2; do
    echo "--- variant $v chunk $c" >> $LOG
    M355_DECODE_VARIANT=$v timeout -k 10 120 python -u bench.py --steps 5 --warmup 2 --no-cpu-baseline --chunk $c 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['us_per_launch'], d['roofline']['achieved'], d['ms_per_step'])" >> $LOG 2>&1
  done
done
export TMPDIR=/tmp
for v in $3; do
  export M355_DECODE_VARIANT=$v
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_v$v -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --chunk 256 > gpurun_out/prof_v$v.log 2>&1
  echo "=== rocprof variant $v" >> $LOG
  find gpurun_out/prof_v$v -name "*kernel_stats.csv" -exec cut -c1-160 {} \; >> $LOG 2>&1
done
grep -E "^===|^---|^[0-9]|FAILED|passed|failed|\"" $LOG
