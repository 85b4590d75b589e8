#!/bin/bash
# tools/hp_factor_ab.sh: the kernels of the half-product form back to back (tools/bench_kernels.py --reorder auto, 100^3) with library variants
# build_variants/libopmhip_NAME.so in alternation inside one GPU session
for round in 1 2; do
for V in "$@"; do
  echo "== $V (round $round)"
  OPMHIP_LIB=build_variants/libopmhip_$V.so python tools/bench_kernels.py --n 100 --reorder auto --reps 30 2>&1 | grep -E "ms "
done
done
