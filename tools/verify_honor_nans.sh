#!/bin/bash
# The library is built with -fno-honor-nans (sslap_amd/build.py: it only removes NaN-quieting copies).
# This script shows that the flag changes no result: the same sources built WITHOUT it give the same bits on the whole
# -m gpu parity suite (every fixture, the round-by-round traces, the BASELINE hashes).
#   bash tools/verify_honor_nans.sh build       (build container: build_ab/lib_honor_nans.so)
#   bash tools/verify_honor_nans.sh run [out]   (GPU box: the suite on that build -> gpurun_out/<out>/honor_nans_suite.txt)
R=$(cd "$(dirname "$0")/.." && pwd); cd "$R"
if [ "$1" = build ]; then
  mkdir -p build_ab
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -shared -fPIC -fvisibility=hidden \
    sslap_amd/csrc/misslap.hip -o build_ab/lib_honor_nans.so && echo "built build_ab/lib_honor_nans.so (no -fno-honor-nans)"
  exit $?
fi
O=gpurun_out/${2:-verify}; mkdir -p "$O"
{ echo "# -m gpu parity suite on build_ab/lib_honor_nans.so (the product sources built WITHOUT -fno-honor-nans)";
  echo "# library: $(sha256sum build_ab/lib_honor_nans.so | cut -c1-16)  product build: $(sha256sum sslap_amd/libmisslap.so | cut -c1-16)";
} > "$O/honor_nans_suite.txt"
MISSLAP_LIB=$R/build_ab/lib_honor_nans.so timeout -k 10 900 python -X faulthandler -m pytest tests/test_gpu_parity.py tests/test_matching.py -m gpu -x -q >> "$O/honor_nans_suite.txt" 2>&1 || { tail -30 "$O/honor_nans_suite.txt"; exit 1; }
tail -3 "$O/honor_nans_suite.txt"
