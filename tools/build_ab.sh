#!/bin/bash
# Build variants of the library for an A/B on one GPU box: bash tools/build_ab.sh <name> "<extra hipcc flags>" ...
# -> build_ab/lib_<name>.so (selected at run time with MISSLAP_LIB; tools/ab_bench.sh / tools/ab_libs.sh run them).
set -e
R=$(cd "$(dirname "$0")/.." && pwd); mkdir -p "$R/build_ab"
while [ $# -ge 2 ]; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-honor-nans -std=c++17 -shared -fPIC -fvisibility=hidden $2 \
    "$R/sslap_amd/csrc/misslap.hip" -o "$R/build_ab/lib_$1.so" && echo "built lib_$1.so ($2)"
  shift 2
done
