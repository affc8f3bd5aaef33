#!/bin/bash
# A/B of a macro-selected kernel variant against the shipped build, both in ONE gpurun call (boxes differ by ~0.5 %):
#   scripts/ab_variant.sh build "-DIIF_NT_RES" conv_igemm        (here, cross-compiles _debug/libiif_amd_var.so)
#   scripts/ab_variant.sh run [bench args]                        (on the GPU box: A B A B, ms per step)
set -e
cd "$(dirname "$0")/../iif_amd/csrc"
if [ "$1" = build ]; then
    mkdir -p _debug
    objs=""
    for f in bn conv_igemm conv_wgrad elementwise fasa iif_head mask_head norm_head pool_misc se; do
        if [[ " ${@:3} " == *" $f "* ]]; then
            /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I../../include $2 -c $f.hip -o _debug/${f}_var.o
            objs="$objs _debug/${f}_var.o"
        else
            objs="$objs $f.o"
        fi
    done
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o _debug/libiif_amd_var.so $objs iif_host.o
    echo built _debug/libiif_amd_var.so
else
    shift
    cd ../..
    mkdir -p gpurun_out/ab
    for i in 1 2; do
        timeout -k 10 250 python bench.py --steps 30 --warmup 5 --no-cpu-baseline "$@" > gpurun_out/ab/a$i.log 2>&1
        IIF_AMD_LIB=$PWD/iif_amd/csrc/_debug/libiif_amd_var.so timeout -k 10 250 python bench.py --steps 30 --warmup 5 --no-cpu-baseline "$@" > gpurun_out/ab/b$i.log 2>&1
    done
    echo "shipped: $(grep -h -o '"ms_per_step": [0-9.]*' gpurun_out/ab/a?.log | tr '\n' ' ')"
    echo "variant: $(grep -h -o '"ms_per_step": [0-9.]*' gpurun_out/ab/b?.log | tr '\n' ' ')"
fi
