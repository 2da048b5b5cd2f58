# bench_one_batch on one GPU -> gpurun_out/obb/*.jsonl (copied into profiles/ by hand): eager and graphed decode
set -e
cd /root/repo
mkdir -p gpurun_out/obb
for g in "" "--graph"; do
  sfx=${g:+_graph}
  echo "8b $g"; timeout -k 10 400 python mini-sglang_amd/bench_one_batch.py --model llama-3-8b --batch-size 1 16 64 --input-len 1024 --output-len 33 --page-size 16 $g --result-filename gpurun_out/obb/r01_bench_one_batch_8b_tp1$sfx.jsonl > gpurun_out/obb/8b$sfx.log 2>&1
  echo "1b $g"; timeout -k 10 300 python mini-sglang_amd/bench_one_batch.py --model llama-3.2-1b --batch-size 1 64 --input-len 1024 --output-len 33 --page-size 16 $g --result-filename gpurun_out/obb/r01_bench_one_batch_1b_tp1$sfx.jsonl > gpurun_out/obb/1b$sfx.log 2>&1
done
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/obb/*.jsonl")):
    for l in open(f):
        d = json.loads(l)
        print(f.split("/")[-1], d["batch_size"], round(d["prefill_throughput"]), round(d["median_decode_latency"] * 1e3, 3), round(d["median_decode_throughput"]))
PY
