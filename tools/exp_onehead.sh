#!/bin/bash
# Dev probe: decode kernel time with only one kv head streaming (M355_STREAM_ROT=100+h), to see whether a head's
# slowness follows its addresses.
export TMPDIR=/tmp
for h in "$@"; do
  rm -rf /tmp/pp
  M355_STREAM_ROT=$h timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /tmp/l.log 2>&1 || { tail -3 /tmp/l.log; exit 1; }
  python3 - "$h" <<'PY'
import csv, glob, sys
f = glob.glob("/tmp/pp/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "decode_mfma" in r["Name"] or "decode_stream_kernel" in r["Name"]:
        print("rot", sys.argv[1], "avg_us", round(float(r["AverageNs"]) / 1e3, 2), "min", round(float(r["MinNs"]) / 1e3, 2), "max", round(float(r["MaxNs"]) / 1e3, 2))
PY
done
