export TMPDIR=/tmp
mkdir -p gpurun_out/turnprof
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/turnprof -o t -- python3 tools/bench_prefill.py --prefix 8192 --suffix 64,64 --groups 1 --per-group 1 --layers 4 --reps 5 > gpurun_out/turnprof/log.txt 2>&1
python3 - <<'PY'
import sqlite3, glob
db = glob.glob("gpurun_out/turnprof/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
for n, cnt, av in c.execute(f"select s.kernel_name, count(*), avg(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id where s.kernel_name like '%m355%' group by s.kernel_name"):
    print(f"n={cnt} avg {av/1e3:.2f} us {n[:90]}")
rows = c.execute(f"select d.start, d.end, s.kernel_name from {kd} d join {ks} s on d.kernel_id=s.id where s.kernel_name like '%m355%' order by d.start").fetchall()
gaps = [(rows[i+1][0]-rows[i][1])/1e3 for i in range(len(rows)-1) if 'merge' in rows[i+1][2]]
print("gap before merge (us): median", sorted(gaps)[len(gaps)//2] if gaps else None)
PY
