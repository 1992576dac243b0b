#!/bin/bash
# GPU box: run a script with an experiment build of the library swapped in.  usage: exp_variant.sh <variant> <cmd...>
v=$1; shift
cp pathtracer_amd/libptcore.so /tmp/libptcore_keep.so
cp pathtracer_amd/libptcore_exp_$v.so pathtracer_amd/libptcore.so
"$@"
rc=$?
cp /tmp/libptcore_keep.so pathtracer_amd/libptcore.so
exit $rc
