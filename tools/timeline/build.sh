#!/bin/bash
# measurement build of the library with the step kernels' timeline stamps (-DPX_TIMELINE); never the product library
set -e
cd "$(dirname "$0")/../../proxsdp.jl_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I/opt/rocm/include -Wall -Wno-unused-function \
    -Xarch_host -mavx2 -Xarch_host -ffp-contract=off -pthread -DPX_TIMELINE capi.hip \
    -o ../../tools/timeline/libproxsdp_hip_tl.so -shared -L/opt/rocm/lib -lrocsolver -lrocblas -Wl,-rpath,/opt/rocm/lib
