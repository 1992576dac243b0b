#!/bin/bash
# Builds pathtracer_amd/libptcore_exp_<name>.so: the library with extra compiler flags on the HIP source (the host-only objects
# of the last product build are reused).  usage: build_variant.sh <name> <flags...>
set -e
cd "$(dirname "$0")/../pathtracer_amd/csrc"
n=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fvisibility=hidden -pthread -Wall -Wno-unused-function -Wno-unused-result "$@" -c ptcore.hip -o /tmp/ptcore_exp_$n.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libptcore_exp_$n.so /tmp/ptcore_exp_$n.o bvh_build.cpp.o ply.cpp.o scene_prep.cpp.o frame_dump.cpp.o -ldl -lz -pthread
