#!/bin/bash
# Dev: (waves per CU, ring depth) grid of the MFMA decode kernel on the bench workload; args: "wpc,ring[,rot]" ...
export TMPDIR=/tmp
for cfg in "$@"; do
  IFS=, read wpc ring rot <<< "$cfg"
  rm -rf /tmp/pp
  M355_MFMA_WPC=$wpc M355_MFMA_RING=$ring M355_STREAM_ROT=${rot:-0} timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /tmp/l.log 2>&1 || { tail -3 /tmp/l.log; exit 1; }
  python3 - "$cfg" <<'PY'
import csv, glob, sys
f = glob.glob("/tmp/pp/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "decode_mfma" in r["Name"] or "merge" in r["Name"]:
        print("cfg", sys.argv[1], r["Name"][:40], "avg_us", round(float(r["AverageNs"]) / 1e3, 2), "min", round(float(r["MinNs"]) / 1e3, 2))
PY
done
