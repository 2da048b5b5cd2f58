#!/bin/bash
# Dev: loads-only / compute-only / full variants of the chunked decode kernel on the bench workload.
export TMPDIR=/tmp M355_DECODE_MODE=chunked
for v in 0 11 12 13; do
  export M355_DECODE_VARIANT=$v
  rm -rf /tmp/abl_$v
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abl_$v -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /tmp/abl.log 2>&1 || { tail -5 /tmp/abl.log; exit 1; }
  echo "=== variant $v"
  find /tmp/abl_$v -name "*kernel_stats.csv" -exec grep -E "decode_" {} \; | cut -c1-150
done
