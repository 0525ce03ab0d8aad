#!/bin/bash
# AddressSanitizer on the HOST code of the library (formatter, scene generation, hierarchy builder, error paths); no GPU needed.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $ROOT/build/asan
cd $ROOT/raytracing_amd/csrc
/opt/rocm/bin/hipcc -O1 -g -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared -I$ROOT/include -Xarch_host -fsanitize=address \
    -o $ROOT/build/asan/librt_amd.so rt_api.hip rt_trace.hip rt_wavefront.hip rt_pixel.hip rt_ppm.hip rt_scene.hip rt_unit.hip
/opt/rocm/lib/llvm/bin/clang++ -O1 -g -std=c++17 -fsanitize=address -I$ROOT/include -o $ROOT/build/asan/driver $ROOT/tools/asan_driver.cpp \
    -L$ROOT/build/asan -lrt_amd -Wl,-rpath,$ROOT/build/asan -Wl,-rpath,/opt/rocm/lib
ASAN_OPTIONS=detect_leaks=0 $ROOT/build/asan/driver
