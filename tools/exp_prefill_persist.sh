cd /root/repo
# A/B: persistent row-block loop + KV-split last round (product library) against the one-block-per-workgroup form (lib/libmi355attn_pbase.so)
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q -k "prefill or cfg3 or extend or radix" --timeout 300 --timeout-method thread 2>&1 | tail -3 || exit 1
for i in 1 2; do for lib in pbase ""; do
  L=$PWD/mini-sglang_amd/lib/libmi355attn${lib:+_$lib}.so
  for a in "" "--groups 8" "--heads 8,1,128" "--prefix 0 --groups 1 --per-group 4 --suffix 2048,2048 --layers 2 --reps 3" "--prefix 0 --groups 2 --per-group 1 --suffix 8192,8192 --layers 2 --reps 3"; do
    echo -n "${lib:-new} [$a] "
    M355_LIB_PATH=$L timeout -k 10 200 python tools/bench_prefill.py $a 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['us_per_layer'], d['tflops'])" || exit 1
  done
done; done
echo "=== persist=1, tail split, cfg3"; M355_PREFILL_PERSIST=1 timeout -k 10 200 python tools/stamp_prefill_cfg3.py 2>&1 | grep -v amdgpu.ids
