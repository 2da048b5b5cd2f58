#!/bin/bash
# Dev: SQ / TCC counters of the decode kernel; args are passed to bench.py
export TMPDIR=/tmp
rm -rf /tmp/pmc2
timeout -s KILL 180 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d /tmp/pmc2 -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" > /tmp/pmc2.log 2>&1 || { tail -5 /tmp/pmc2.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/pmc2/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "decode_mfma" not in k: continue
    acc[k[:50]][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k[:50], r["Counter_Name"])] += 1
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()): print(f"   {c:28s} {v / n[(k, c)]:14.0f} per launch")
PY
