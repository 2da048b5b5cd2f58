cd /root/repo
for m in "M355_DECODE_MODE=chunked" "M355_STREAM_KERNEL=valu"; do
  echo -n "$m "
  env $m timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['us_per_launch'], d['roofline']['frac'])" || exit 1
done
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -x -q -k "decode or sweep or flow" --timeout 120 --timeout-method thread 2>&1 | tail -1
