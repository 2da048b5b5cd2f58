#!/bin/bash
# rocprofv3 kernel trace of the small-launch decode shapes (range kernel + merge): how the per-layer time splits between the
# two launches.  Durations come from the rocpd database rocprofv3 writes; the profiler serialises launches, so the sum of the two
# kernels is above the HIP-event time of the unprofiled run (printed first).
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_small
i=0
for v in "--heads 8,1,128" "--heads 8,1,128 --fixed-len 32768 --batch 1" "--heads 8,1,128 --fixed-len 32768 --batch 8" "--heads 8,2,128"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_small/$i -o s -- python3 bench.py --no-cpu-baseline --no-extras --no-pmc --steps 10 --warmup 3 $v > gpurun_out/prof_small/$i.log 2>&1 || exit 1
  echo "== $v  (under the profiler: $(grep -o '"us_per_launch": [0-9.]*' gpurun_out/prof_small/$i.log | tail -1))"
  python3 - <<P
import sqlite3, statistics
db = sqlite3.connect('gpurun_out/prof_small/$i/s_results.db')
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]; ks = [t for t in tabs if 'kernel_symbol' in t][0]
q = f"select s.kernel_name, count(*), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name"
for r in db.execute(q):
    if 'decode_mfma' in r[0] or 'stream_merge' in r[0]:
        print('  ', r[0][9:40], 'calls', r[1], 'avg ns', round(r[2]), 'min', r[3], 'max', r[4])
P
done
