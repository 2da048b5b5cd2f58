#!/bin/bash
# Timing ablations of the dev-only 64-rows-per-wave prefill kernel (csrc/prefill_wg2.hip).
#   tools/wg2_ablation.sh build     here (CPU): one dev library per -DM355_WG2_EXPERIMENT=n under mini-sglang_amd/lib/exp/
#   tools/wg2_ablation.sh run       on the GPU box: 2 x 8192 prompt through each, lines into gpurun_out/wg2_ablation.jsonl
#   tools/wg2_ablation.sh clean
# The ablated builds compute WRONG results on purpose (they leave work out); they exist to price the parts of the loop.
set -e
cd "$(dirname "$0")/.."
CS=mini-sglang_amd/csrc
EXP=mini-sglang_amd/lib/exp
case "$1" in
build)
    make -C $CS dev -j8 > /dev/null
    mkdir -p $EXP /tmp/wg2_abl
    for n in 1 2 3 4 5; do
        /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form=1 \
            -DM355_DEV -DM355_WG2_EXPERIMENT=$n -c $CS/prefill_wg2.hip -o /tmp/wg2_abl/wg2_$n.o &
    done
    wait
    for n in 1 2 3 4 5; do
        /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $(ls $CS/build/dev/*.o | grep -v prefill_wg2) /tmp/wg2_abl/wg2_$n.o -o $EXP/libwg2_exp$n.so
    done
    ;;
run)
    mkdir -p gpurun_out
    : > gpurun_out/wg2_ablation.jsonl
    names=("product kernel (prefill_wg_kernel)" "wg2 full" "wg2 without steady-state LDS-DMA" "wg2 without the pair barrier" "wg2 without exp2/convert/sum chunks" "wg2 without DMA and softmax chunks" "wg2 bare MFMA stream (no DMA, softmax, LDS fragment reads)")
    for n in p 0 1 2 3 4 5; do
        if [ $n = p ]; then lib=dev; rt=1; i=0; elif [ $n = 0 ]; then lib=dev; rt=2; i=1; else lib=$PWD/$EXP/libwg2_exp$n.so; rt=2; i=$((n + 1)); fi
        M355_LIB_PATH=$lib M355_PREFILL_RT=$rt timeout -k 10 120 python tools/bench_prefill.py --prefix 0 --suffix 8192,8192 --groups 1 \
            --per-group 2 --layers 2 --reps 3 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); d['build']='${names[$i]}'; print(json.dumps(d))" >> gpurun_out/wg2_ablation.jsonl
    done
    cat gpurun_out/wg2_ablation.jsonl
    ;;
clean) rm -rf $EXP /tmp/wg2_abl ;;
*) echo "usage: $0 build|run|clean"; exit 2 ;;
esac
