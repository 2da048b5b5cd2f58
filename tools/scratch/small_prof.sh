export TMPDIR=/tmp
for v in "--heads 8,1,128" "--heads 8,1,128 --fixed-len 32768 --batch 8" "--heads 8,1,128 --fixed-len 32768 --batch 1"; do
rm -rf gpurun_out/smallprof; mkdir -p gpurun_out/smallprof
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/smallprof -o t -- python3 bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 2 $v > gpurun_out/smallprof/log.txt 2>&1
echo "== $v"
python3 - <<'PY'
import sqlite3, glob
db = glob.glob("gpurun_out/smallprof/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
for n, cnt, av in c.execute(f"select s.kernel_name, count(*), avg(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id where s.kernel_name like '%m355%' group by s.kernel_name order by 3 desc"):
    print(f"n={cnt} avg {av/1e3:.2f} us {n[:100]}")
PY
done
