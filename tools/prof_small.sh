#!/bin/bash
# rocprofv3 kernel stats of the small-launch decode shapes (range kernel + merge): how the per-layer time splits between the two launches
export TMPDIR=/tmp
i=0
for v in "--heads 8,1,128" "--heads 8,1,128 --fixed-len 32768 --batch 1" "--heads 8,1,128 --fixed-len 32768 --batch 8" "--heads 8,2,128"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_small/$i -o s -- python3 bench.py --no-cpu-baseline --no-extras --no-pmc --steps 10 --warmup 3 $v > gpurun_out/prof_small/$i.log 2>&1 || exit 1
  echo "== $v"; tail -1 gpurun_out/prof_small/$i.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('us/launch', d['roofline']['us_per_launch'], 'ms/step', d['ms_per_step'])"
  python3 - <<P
import csv,glob
f=glob.glob('gpurun_out/prof_small/$i/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'm355' in r['Name']: print('  ', r['Name'][:60], 'calls', r['Calls'], 'avg ns', r['AverageNs'], 'min', r['MinNs'], 'max', r['MaxNs'])
P
done
