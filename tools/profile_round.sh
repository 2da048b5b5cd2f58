#!/bin/bash
# Round profile of the exact default bench.py command (run on the GPU box through gpurun):
#   1. rocprofv3 --kernel-trace --stats           -> gpurun_out/prof_<tag>/stats
#   2. rocprofv3 --pmc FETCH_SIZE  (own pass)     -> gpurun_out/prof_<tag>/fetch
#   3. rocprofv3 --pmc WRITE_SIZE  (own pass)     -> gpurun_out/prof_<tag>/write
# then tools/summarize_profile.py turns the three into profiles/<tag>_*.{csv,json} (run here, on the CPU side).
set -e
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 bench.py --no-cpu-baseline --no-extras"      # the headline workload alone (the extras run other kernels / shapes)
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- $BENCH > $OUT/stats.log 2>&1
echo "stats pass done"
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o fetch -- $BENCH --steps 4 --warmup 2 > $OUT/fetch.log 2>&1
echo "fetch pass done"
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o write -- $BENCH --steps 4 --warmup 2 > $OUT/write.log 2>&1
echo "write pass done"
timeout -k 10 240 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json
#   4. the prefill (extend) kernel on BASELINE configs[2] (tools/bench_prefill.py): kernel stats + the tool's own line
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $OUT/prefill -o prefill -- python3 tools/bench_prefill.py --layers 8 --reps 5 > $OUT/prefill.log 2>&1
echo "prefill stats pass done"
timeout -k 10 240 python3 tools/bench_prefill.py > $OUT/prefill.json 2> $OUT/prefill.err
cat $OUT/prefill.json
#   5. MFMA utilisation of the prefill kernel: one --pmc pass (5 SQ + 1 GRBM counter)
timeout -s KILL 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d $OUT/prefill_pmc -o pmc -- python3 tools/bench_prefill.py --layers 2 --reps 2 > $OUT/prefill_pmc.log 2>&1
echo "prefill pmc pass done"
