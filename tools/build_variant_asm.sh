#!/bin/bash
# tools/build_variant_asm.sh NAME "FLAGS": like build_variant.sh, but assemble.hip is the file compiled under the flags
set -e
cd "$(dirname "$0")/../opm-autodiff_amd"
mkdir -p ../build_variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math $2 -c csrc/assemble.hip -o ../build_variants/assemble_$1.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../build_variants/libopmhip_$1.so ../build_variants/assemble_$1.o csrc/comm.o csrc/cpr.o csrc/solver.o csrc/capi.o csrc/capi_asm.o csrc/reorder.o csrc/fluid_tables.o -ldl -lpthread
rm -f ../build_variants/assemble_$1.o
