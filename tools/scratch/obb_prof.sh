set -e
export TMPDIR=/tmp
mkdir -p gpurun_out/obbprof
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/obbprof -o obb -- python3 mini-sglang_amd/bench_one_batch.py --model llama-3-8b --batch-size 64 --input-len 1024 --output-len 17 --page-size 16 > gpurun_out/obbprof/log.txt 2>&1
python3 - <<'PY'
import sqlite3, glob
db = glob.glob("gpurun_out/obbprof/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = c.execute(f"select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
for n, cnt, sm, av in rows[:28]:
    print(f"{sm/1e6:9.2f} ms {100*sm/tot:5.1f}%  n={cnt:6d} avg {av/1e3:9.2f} us  {n[:110]}")
PY
