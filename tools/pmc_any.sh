#!/bin/bash
# Dev: one --pmc pass over a python tool.  usage: pmc_any.sh "<counters>" <kernel-substring> <script> [args...]
# (<= 8 SQ_, 4 TCC_ counters per pass; FETCH_SIZE counts as 3 TCC, WRITE_SIZE as 2)
export TMPDIR=/tmp
ctr="$1"; pat="$2"; shift 2
rm -rf /tmp/pmc_any
timeout -s KILL 150 rocprofv3 --pmc $ctr --output-format csv -d /tmp/pmc_any -- python3 "$@" > /tmp/pmc_any.log 2>&1 || { tail -5 /tmp/pmc_any.log; exit 1; }
PAT="$pat" python3 - <<'PY'
import csv, glob, collections, os
f = glob.glob("/tmp/pmc_any/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if os.environ["PAT"] not in k: continue
    acc[k[:60]][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k[:60], r["Counter_Name"])] += 1
for k, d in acc.items():
    print(k)
    for c, v in d.items(): print(f"   {c:28s} {v / n[(k, c)]:16.0f} per launch")
PY
