# A/B of two library builds on the prefill workloads (one box): tools/ab_prefill_lib.sh libA.so libB.so
cd /root/repo
for i in 1 2; do for lib in "$@"; do
  for a in "" "--prefix 0 --groups 2 --per-group 1 --suffix 8192,8192" "--heads 8,1,128 --prefix 0 --groups 1 --per-group 1 --suffix 32768,32768 --layers 2 --reps 3" "--heads 32,8,64"; do
    echo -n "$(basename $lib) [$a] "
    M355_LIB_PATH=$lib timeout -k 10 200 python tools/bench_prefill.py $a 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['us_per_layer'], d['tflops'])" || exit 1
  done
done; done
