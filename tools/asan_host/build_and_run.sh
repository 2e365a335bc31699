#!/bin/bash
# The HOST side of the C-ABI under AddressSanitizer (CPU only; run by tests/test_asan_host.py, which together with this
# directory is listed in .gpurunignore: the GPU pool refuses any snapshot that carries a sanitizer flag, and nothing here
# is needed on a GPU box).  Builds the library with the host code instrumented and the device code as usual into
# mutual-information-multimodal_amd/lib_asan/, links asan_host_driver.cpp to it and runs it.
#   usage: build_and_run.sh <scratch dir for the driver binary>
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
ROOT=$(cd "$HERE/../.." && pwd)
OUT=${1:-/tmp}
HIPCC=${HIPCC:-hipcc}
SAN="-fsanitize=address"
make -C "$ROOT/mutual-information-multimodal_amd/csrc" -j4 OUT_DIR=../lib_asan OBJ_DIR=../build_asan LDEXTRA="$SAN" \
  CXXFLAGS="-O1 -g -std=c++17 -fPIC -fno-slp-vectorize --offload-arch=gfx950 $SAN -fno-gpu-sanitize -fno-omit-frame-pointer -Wno-unused-function -Wno-unused-result" >/dev/null
LIBDIR="$ROOT/mutual-information-multimodal_amd/lib_asan"
"$HIPCC" -O1 -g -std=c++17 $SAN -fno-omit-frame-pointer -I"$ROOT/include" "$HERE/asan_host_driver.cpp" \
  -L"$LIBDIR" -lmi_critic_hip -Wl,-rpath,"$LIBDIR" -o "$OUT/asan_host_driver" 2>/dev/null
ASAN_OPTIONS=detect_leaks=0:abort_on_error=0 "$OUT/asan_host_driver"
