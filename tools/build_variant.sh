#!/bin/bash
# tools/build_variant.sh NAME "FLAGS": libopmhip with solver.hip compiled under extra -D flags -> build_variants/libopmhip_NAME.so
# (kernel tuning experiments; select at run time with OPMHIP_LIB=build_variants/libopmhip_NAME.so)
set -e
cd "$(dirname "$0")/../opm-autodiff_amd"
mkdir -p ../build_variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math $2 -c csrc/solver.hip -o ../build_variants/solver_$1.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../build_variants/libopmhip_$1.so ../build_variants/solver_$1.o csrc/comm.o csrc/cpr.o csrc/assemble.o csrc/capi.o csrc/capi_asm.o csrc/reorder.o csrc/fluid_tables.o -ldl -lpthread
rm -f ../build_variants/solver_$1.o
