#!/bin/bash
# A/B of library builds on one box: bench.py headline (no extras) for each lib/libmi355attn_<tag>.so given, twice each, interleaved.
# usage: tools/ab_libs.sh tag1 tag2 ...   ("base" = the product library)
set -e
mkdir -p gpurun_out
for rep in 1 2; do
  for tag in "$@"; do
    lib=mini-sglang_amd/lib/libmi355attn_$tag.so
    [ "$tag" = base ] && lib=mini-sglang_amd/lib/libmi355attn.so
    M355_LIB_PATH=$PWD/$lib timeout -k 10 120 python bench.py --no-extras --no-cpu-baseline ${AB_ARGS} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$tag', d['roofline']['us_per_launch'], d['roofline']['frac'], d['ms_per_step'])"
  done
done
