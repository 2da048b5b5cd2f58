#!/bin/bash
# Decode-attention launch timings over the workloads DESIGN 3.3 tabulates -> one JSON line each (profiles/rNN_decode_variants.jsonl).
# usage (on the GPU box, from the repo root): tools/run_dec_variants.sh > gpurun_out/decode_variants.jsonl
for a in "" "--len-scale 8" "--copies 8" "--heads 32,8,128" "--heads 8,2,128" "--heads 8,1,128" \
         "--heads 8,1,128 --fixed-len 32768 --batch 32" "--heads 8,1,128 --fixed-len 32768 --batch 8" \
         "--heads 8,1,128 --fixed-len 32768 --batch 1" "--fixed-len 2048 --batch 8"; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 3 $a 2>/dev/null | A="$a" python3 -c "
import sys, json, os
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']; c = d['config']
print(json.dumps({'args': os.environ['A'], 'heads_per_gpu': c['heads_per_gpu'], 'batch': c['global_batch'], 'sum_seq_len': c['sum_seq_len'],
                  'kernel': r['kernel'], 'us_per_launch': r['us_per_launch'], 'GBps': r['achieved'], 'frac': r['frac'], 'ms_per_step': d['ms_per_step']}))"
done
