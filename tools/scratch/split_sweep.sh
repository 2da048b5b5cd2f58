for shape in "--prefix 8192 --suffix 64,64 --groups 1 --per-group 1" "--prefix 32768 --suffix 128,128 --groups 1 --per-group 1" "--prefix 4096 --suffix 16,16 --groups 1 --per-group 4" "--prefix 16384 --suffix 32,32 --groups 2 --per-group 1" "--prefix 2048 --suffix 64,64 --groups 1 --per-group 2"; do
  echo "== $shape"
  for sp in 0 2 4 8 16 32; do
    M355_LIB_PATH=dev M355_PREFILL_SPLITS=$sp timeout -k 10 100 python tools/bench_prefill.py $shape --layers 4 --reps 5 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   splits=$sp', d['us_per_layer'], 'us', d['tflops'], 'TF')" || exit 1
  done
done
