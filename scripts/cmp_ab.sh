#!/bin/bash
# GPU box: scripts/exp_ab.py with the product library ("base...") and with experiment builds, one after the other on the same box.
# usage: cmp_ab.sh <tag> "<exp_ab.py arguments>" <variant>...    results in gpurun_out/<tag>.txt
tag=$1; xargs=$2; shift 2
for v in "$@"; do
    if [ "${v%%[0-9]*}" = base ]; then unset PTCORE_LIB; else export PTCORE_LIB=$PWD/pathtracer_amd/libptcore_exp_$v.so; fi
    timeout -k 10 240 python scripts/exp_ab.py $xargs > gpurun_out/${tag}_$v.log 2>&1 || { echo "$v failed"; tail -5 gpurun_out/${tag}_$v.log; exit 1; }
    echo "$v: $(tail -1 gpurun_out/${tag}_$v.log)" | tee -a gpurun_out/$tag.txt
done
