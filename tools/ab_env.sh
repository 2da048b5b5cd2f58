# A/B of one environment switch on one box: tools/ab_env.sh VAR "v1 v2 ..." [bench args]
cd /root/repo
VAR=$1; VALS=$2; shift 2
for i in 1 2 3; do for v in $VALS; do
  echo -n "$VAR=$v [$*] "
  env $VAR=$v timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --warmup 3 "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['us_per_launch'], d['roofline']['frac'])" || exit 1
done; done
