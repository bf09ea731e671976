#!/bin/bash
# alternating physics-only tick rates of library variants on one box: tools/ab_tick_rate.sh <rounds> <envs> <team size> <name> [<name> ...]   ("tree" = the product library)
R=$1; N=$2; TS=$3; shift 3
for i in $(seq $R); do for n in "$@"; do
  if [ "$n" = tree ]; then unset RLGPU_LIB; else export RLGPU_LIB=rlgymppo_cpp_amd/librlgpu_$n.so; fi
  python tools/tick_rate.py $N 200 $TS 2>&1 | tail -1
done; done
