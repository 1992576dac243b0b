#!/bin/bash
# VGPRs, scratch bytes and wavefronts per SIMD of every kernel in ptcore.hip, one per line, sorted by name (no GPU needed).
# Two registers more can cost a traversal kernel a wavefront per SIMD: diff the output of two trees before trusting a change.
# usage: kernel_resources.sh [csrc directory] [extra compiler flags...]
dir=${1:-$(dirname "$0")/../pathtracer_amd/csrc}; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fvisibility=hidden -pthread -Wno-unused-function \
    -Wno-unused-result "$@" --cuda-device-only -Rpass-analysis=kernel-resource-usage -c "$dir/ptcore.hip" -o /dev/null 2>&1 |
  grep -E "Function Name|VGPRs:|Occupancy|ScratchSize" | sed 's/.*remark: *//; s/\[-Rpass.*//' | paste - - - - |
  awk '{printf "%-70s vgprs %3d scratch %4d waves/simd %d\n", $3, $5, $8, $11}' | sort
