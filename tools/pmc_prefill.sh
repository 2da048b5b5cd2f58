#!/bin/bash
# Dev: SQ counters of the prefill kernel on the cfg3 workload (one pass, 8 SQ counters).
export TMPDIR=/tmp
rm -rf /tmp/pmc1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/pmc1 -- python3 tools/bench_prefill.py --layers 2 --reps 2 "$@" > /tmp/pmc1.log 2>&1 || { tail -5 /tmp/pmc1.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/pmc1/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "prefill" not in k: continue
    acc[k[:60]][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k[:60], r["Counter_Name"])] += 1
for k, d in acc.items():
    print(k)
    for c, v in d.items(): print(f"   {c:28s} {v / n[(k, c)]:14.0f} per launch")
PY
