#!/bin/bash
# Build a variant of the stepper next to the product library for A/B runs (tools/ab_bench.sh):
#   tools/build_variant.sh <name> <extra hipcc flags...>   ->  rlgymppo_cpp_amd/librlgpu_<name>.so  (+ _obj/rlgpu_env_<name>.resource.log)
set -e
NAME=$1; shift
cd "$(dirname "$0")/../rlgymppo_cpp_amd/csrc"
make -s all
# (through the wrapper that repairs the whole-wave-bracket defect, like the product build: DESIGN.md 4.1)
HIPCC=/opt/rocm/bin/hipcc python3 ../../tools/hipcc_wwm_safe.py --log _obj/rlgpu_env_$NAME.wwm.log -std=c++17 -O3 -fPIC --offload-arch=gfx950 -Wno-unused-value -ffp-contract=off -Rpass-analysis=kernel-resource-usage "$@" \
    -c rlgpu_env.hip -o _obj/rlgpu_env_$NAME.o 2> _obj/rlgpu_env_$NAME.resource.log || { cat _obj/rlgpu_env_$NAME.resource.log; exit 1; }
cat _obj/rlgpu_env_$NAME.wwm.log
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 _obj/rlgpu_env_$NAME.o _obj/rlgpu_learn.o _obj/rlgpu_comm.o _obj/arena_mesh.o _obj/lt_archive.o \
    -o ../librlgpu_$NAME.so -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
grep -A10 "k_env_collectILi2" _obj/rlgpu_env_$NAME.resource.log | grep -E "VGPRs:|AGPRs|Scratch|Occupancy|LDS Size" | tr -s ' ' | tr '\n' ';'; echo
