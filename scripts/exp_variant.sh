#!/bin/bash
# GPU box: run a command with an experiment build of the library loaded instead of the product library.  The product
# file is never touched: pathtracer_amd.core.lib_path() honours PTCORE_LIB.  usage: exp_variant.sh <variant> <cmd...>
v=$1; shift
PTCORE_LIB=$PWD/pathtracer_amd/libptcore_exp_$v.so exec "$@"
