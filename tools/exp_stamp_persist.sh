cd /root/repo
for a in "" "--prefix 0 --groups 1 --per-group 4 --suffix 2048,2048"; do for m in 0 1; do
  echo "=== persist=$m [$a]"; M355_PREFILL_PERSIST=$m timeout -k 10 200 python tools/stamp_prefill_cfg3.py $a 2>&1 | grep -v amdgpu.ids || exit 1
done; done
