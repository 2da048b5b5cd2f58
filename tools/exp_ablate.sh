mkdir -p gpurun_out; LOG=gpurun_out/ablate.log; : > $LOG
export TMPDIR=/tmp
for cfg in "15 1" "16 1" "14 1"; do
  set -- $cfg
  export M355_DECODE_VARIANT=$1
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_b$1_$2 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --chunk 256 --copies $2 > gpurun_out/prof_b.log 2>&1
  echo "=== variant $1 copies $2" >> $LOG
  find gpurun_out/prof_b$1_$2 -name "*kernel_stats.csv" -exec grep -E "decode_" {} \; | cut -c1-140 >> $LOG 2>&1
done
cat $LOG
