#!/bin/bash
# GPU box: spp/s of scripts/exp_render.py with the product library and with experiment builds, on the same box.
# usage: cmp_frames.sh <tag> "<exp_render.py arguments>" <variant>...   ("base..." names the product library)
tag=$1; rargs=$2; shift 2
for v in "$@"; do
    if [ "${v%%[0-9]*}" = base ]; then unset PTCORE_LIB; else export PTCORE_LIB=$PWD/pathtracer_amd/libptcore_exp_$v.so; fi
    timeout -k 10 200 python scripts/exp_render.py $rargs > gpurun_out/${tag}_$v.log 2>&1 || { echo "$v failed"; tail -3 gpurun_out/${tag}_$v.log; exit 1; }
    echo "$v: $(grep -o '[0-9.]* spp/s.*Mrays/s' gpurun_out/${tag}_$v.log)"
done
