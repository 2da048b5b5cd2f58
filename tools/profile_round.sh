#!/bin/bash
# Round profile (run on the GPU box through gpurun; tools/summarize_profile.py then writes profiles/<tag>_* here):
#   1. rocprofv3 --kernel-trace --stats of the headline workload   -> gpurun_out/prof_<tag>/stats
#   2. rocprofv3 --pmc FETCH_SIZE  (own pass)                      -> gpurun_out/prof_<tag>/fetch
#   3. rocprofv3 --pmc WRITE_SIZE  (own pass)                      -> gpurun_out/prof_<tag>/write
#   4. the full default bench.py line (extras included)            -> gpurun_out/prof_<tag>/bench.json
#   5. the decode variants quoted in DESIGN.md, one line each      -> gpurun_out/prof_<tag>/variants.jsonl
#   6. prefill (tools/bench_prefill.py): kernel stats, its own lines, one --pmc pass (MFMA utilisation)
set -e
TAG=${1:-r02}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 bench.py --no-cpu-baseline --no-extras --no-pmc"      # the headline workload alone (the extras run other kernels / shapes)
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- $BENCH > $OUT/stats.log 2>&1
echo "stats pass done"
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o fetch -- $BENCH --steps 4 --warmup 2 > $OUT/fetch.log 2>&1
echo "fetch pass done"
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o write -- $BENCH --steps 4 --warmup 2 > $OUT/write.log 2>&1
echo "write pass done"
timeout -k 10 400 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench line done"
: > $OUT/variants.jsonl
for v in "--len-scale 8" "--copies 8" "--heads 32,8,128" "--heads 8,2,128" "--heads 8,1,128" "--heads 8,1,128 --fixed-len 32768 --batch 32" \
         "--heads 8,1,128 --fixed-len 32768 --batch 8" "--heads 8,1,128 --fixed-len 32768 --batch 1" "--fixed-len 2048 --batch 8"; do
  timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 $v 2>/dev/null | tail -1 | \
    python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']; print(json.dumps({'args': '$v', 'heads_per_gpu': c['heads_per_gpu'], 'batch': c['global_batch'], 'sum_seq_len': c['sum_seq_len'], 'kernel': r['kernel'], 'us_per_launch': r['us_per_launch'], 'GBps': r['achieved'], 'frac': r['frac'], 'ms_per_step': d['ms_per_step']}))" >> $OUT/variants.jsonl
done
echo "variants done"
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $OUT/prefill -o prefill -- python3 tools/bench_prefill.py --layers 8 --reps 5 > $OUT/prefill.log 2>&1
echo "prefill stats pass done"
timeout -k 10 240 python3 tools/bench_prefill.py > $OUT/prefill.json 2> $OUT/prefill.err
for v in "--heads 32,8,64" "--heads 8,1,128" "--prefix 0 --suffix 2048,2048 --groups 1 --per-group 4 --layers 2 --reps 3" \
         "--prefix 0 --suffix 8192,8192 --groups 1 --per-group 2 --layers 2 --reps 3" \
         "--heads 8,1,128 --prefix 0 --suffix 32768,32768 --groups 1 --per-group 1 --layers 2 --reps 3"; do
  timeout -k 10 240 python3 tools/bench_prefill.py $v >> $OUT/prefill.json 2>> $OUT/prefill.err
done
cat $OUT/prefill.json
timeout -s KILL 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d $OUT/prefill_pmc -o pmc -- python3 tools/bench_prefill.py --layers 2 --reps 2 > $OUT/prefill_pmc.log 2>&1
echo "prefill pmc pass done"
