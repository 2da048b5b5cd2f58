cd /root/repo
bash tools/ab_prefill_lib.sh $PWD/mini-sglang_amd/lib/libmi355attn_pbase.so $PWD/mini-sglang_amd/lib/libmi355attn.so || exit 1
for o in "" desc asc; do for i in 1 2; do
  echo -n "order=[$o] "; M355_EXP_REQ_ORDER=$o timeout -k 10 200 python tools/bench_prefill.py 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['us_per_layer'], d['tflops'])" || exit 1
done; done
