#!/usr/bin/env python3
"""Turn the rocprofv3 outputs of tools/profile_round.sh (rocpd sqlite files under gpurun_out/prof_<tag>/) into
the committed summaries under profiles/:

  profiles/<tag>_bench_kernel_stats.csv   per-kernel calls / total / average duration of `python3 bench.py`
  profiles/<tag>_pmc_traffic.json         per-kernel average HBM bytes per launch from the FETCH_SIZE and
                                          WRITE_SIZE passes (separate runs), corrected as
                                          MI355X_MICROARCH.md "HBM" prescribes: both counters are in KiB;
                                          on gfx950 FETCH_SIZE tallies 128-B requests at 64 B, so reads = 2 x.
bench.py reads the json to fill roofline.traffic.
"""
import csv
import json
import os
import sqlite3
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src = os.path.join(REPO, "gpurun_out", f"prof_{tag}")
dst = os.path.join(REPO, "profiles")
os.makedirs(dst, exist_ok=True)


def short(name: str) -> str:
    name = name.replace("void ", "")
    return name if len(name) < 160 else name[:157] + "..."


db = sqlite3.connect(os.path.join(src, "stats", "stats_results.db"))
rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
                  "group by name order by sum(duration) desc").fetchall()
total = sum(r[2] for r in rows)
with open(os.path.join(dst, f"{tag}_bench_kernel_stats.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["# rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-extras (durations in ns)"])
    w.writerow(["kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct"])
    for n, c, t, a, lo, hi in rows:
        w.writerow([short(n), c, int(t), round(a, 1), int(lo), int(hi), round(100 * t / total, 2)])

traffic = {}
for leg, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    path = os.path.join(src, leg, f"{leg}_results.db")
    if not os.path.exists(path):
        continue
    db = sqlite3.connect(path)
    q = ("select kernel_name, count(*), avg(value) from counters_collection where counter_name = ? "
         "and kernel_name like '%m355::%' group by kernel_name")
    for name, calls, kib in db.execute(q, (counter,)):
        e = traffic.setdefault(short(name), {})
        e[f"{counter}_launches"] = calls
        e[f"{counter}_KiB_raw_avg"] = round(kib, 2)
for e in traffic.values():
    rd = e.get("FETCH_SIZE_KiB_raw_avg", 0.0) * 1024 * 2       # gfx950: 128-B requests tallied at 64 B
    wr = e.get("WRITE_SIZE_KiB_raw_avg", 0.0) * 1024
    e["hbm_read_bytes_per_launch"] = int(rd)
    e["hbm_write_bytes_per_launch"] = int(wr)
    e["hbm_bytes_per_launch"] = int(rd + wr)
bench = None
bj = os.path.join(src, "bench.json")
if os.path.exists(bj):
    with open(bj) as f:
        txt = f.read().strip()
    if txt:
        bench = json.loads(txt.splitlines()[-1])
out = {"command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (one pass each) -- python3 bench.py --no-cpu-baseline --no-extras "
                  "--steps 4 --warmup 2",
       "correction": "bytes = KiB*1024; reads doubled (gfx950 FETCH_SIZE counts 128-B requests as 64 B)",
       "kernels": traffic, "bench_line_same_box": bench}
with open(os.path.join(dst, f"{tag}_pmc_traffic.json"), "w") as f:
    json.dump(out, f, indent=1)
pdb = os.path.join(src, "prefill", "prefill_results.db")
if os.path.exists(pdb):
    db = sqlite3.connect(pdb)
    prow = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
                      "where name like '%m355::%' group by name order by sum(duration) desc").fetchall()
    line = ""
    pj = os.path.join(src, "prefill.json")
    if os.path.exists(pj):
        with open(pj) as f:
            line = f.read().strip().splitlines()[-1]
    with open(os.path.join(dst, f"{tag}_prefill_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["# rocprofv3 --kernel-trace --stats -- python3 tools/bench_prefill.py --layers 8 --reps 5 (durations in ns)"])
        w.writerow(["# un-profiled line of the same tool on the same box: " + line])
        w.writerow(["kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns"])
        for n, c, t, a, lo, hi in prow:
            w.writerow([short(n), c, int(t), round(a, 1), int(lo), int(hi)])
            print(f"{short(n)[:90]:90s} calls {c:5d} avg {a / 1e3:8.2f} us")
ppmc = os.path.join(src, "prefill_pmc", "pmc_results.db")
if os.path.exists(ppmc):
    db = sqlite3.connect(ppmc)
    vals = {}
    for name, cname, calls, avg in db.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                                              "where kernel_name like '%prefill_wg_kernel%' group by kernel_name, counter_name"):
        vals.setdefault(short(name), {})[cname] = avg
    for kname, v in vals.items():
        # SQ_VALU_MFMA_BUSY_CYCLES: cycles summed over the SIMDs (= 32 x v_mfma_f32_32x32x16_bf16 count);
        # GRBM_GUI_ACTIVE: busy cycles summed over the 8 XCDs (MI355X_MICROARCH.md "DVFS give-back")
        cyc = v.get("GRBM_GUI_ACTIVE", 0.0) / 8
        v["mfma_busy_frac_of_simd_cycles"] = round(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (cyc * 1024), 4) if cyc else None
        v["valu_insts_per_mfma"] = round(v.get("SQ_INSTS_VALU", 0.0) / max(v.get("SQ_INSTS_MFMA", 1.0), 1.0), 2)
    with open(os.path.join(dst, f"{tag}_prefill_pmc.json"), "w") as f:
        json.dump({"command": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES "
                              "GRBM_GUI_ACTIVE -- python3 tools/bench_prefill.py --layers 2 --reps 2 (averages per launch)",
                   "kernels": vals}, f, indent=1)
    print(json.dumps(vals, indent=1))
for name in ("variants.jsonl", "prefill.json"):            # measured lines, copied as they are
    path = os.path.join(src, name)
    if os.path.exists(path):
        with open(path) as f, open(os.path.join(dst, f"{tag}_{'decode_variants.jsonl' if name.startswith('variants') else 'prefill_lines.jsonl'}"), "w") as g:
            g.write(f.read())
if bench is not None:
    with open(os.path.join(dst, f"{tag}_bench_full.json"), "w") as f:
        json.dump(bench, f, indent=1)
for k, e in traffic.items():
    print(k[:90], e.get("hbm_read_bytes_per_launch"), e.get("hbm_write_bytes_per_launch"))
for n, c, t, a, lo, hi in rows[:8]:
    print(f"{short(n)[:90]:90s} calls {c:5d} avg {a / 1e3:8.2f} us  {100 * t / total:5.1f}%")
