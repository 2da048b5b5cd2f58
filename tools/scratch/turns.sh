for a in "--prefix 32768 --suffix 128,128 --groups 1 --per-group 1" "--prefix 8192 --suffix 64,64 --groups 1 --per-group 1" "--prefix 8192 --suffix 256,256 --groups 2 --per-group 2" "--prefix 4096 --suffix 32,32 --groups 4 --per-group 4" "--prefix 16384 --suffix 512,512 --groups 1 --per-group 2" "--prefix 2048 --suffix 16,64 --groups 8 --per-group 4"; do
  timeout -k 10 120 python tools/bench_prefill.py $a --layers 4 --reps 5 || exit 1
done
