#!/bin/bash
# GPU box: scripts/exp_ab.py with the product library (any variant name that starts with "base") and with experiment builds
# (scripts/build_variant.sh), one after the other on the same box.
# usage: cmp_ab.sh <tag> "<exp_ab.py arguments>" <variant>...    results in gpurun_out/<tag>.txt
tag=$1; xargs=$2; shift 2
for v in "$@"; do
    case "$v" in
        base*) unset PTCORE_LIB;;
        *) export PTCORE_LIB=$PWD/pathtracer_amd/libptcore_exp_$v.so;;
    esac
    timeout -k 10 240 python scripts/exp_ab.py $xargs > gpurun_out/${tag}_$v.log 2>&1 || { echo "$v failed"; tail -5 gpurun_out/${tag}_$v.log; exit 1; }
    echo "$v [$xargs]: $(tail -1 gpurun_out/${tag}_$v.log)" | tee -a gpurun_out/$tag.txt
done
