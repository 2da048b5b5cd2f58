set -e
cd /root/repo
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -x -q -k "decode" --timeout 120 --timeout-method thread > gpurun_out/t_dec.log 2>&1 || { tail -30 gpurun_out/t_dec.log; exit 1; }
tail -2 gpurun_out/t_dec.log
for a in "" "--len-scale 8" "--heads 8,1,128 --fixed-len 32768 --batch 32 --layers 16" "--heads 8,1,128 --fixed-len 32768 --batch 8 --layers 16" "--copies 8" "--heads 32,8,128" "--heads 8,1,128"; do
  echo "== $a"
  timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --warmup 3 $a 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['us_per_launch'], d['roofline']['frac'])"
done
