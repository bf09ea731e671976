#!/bin/bash
# Build a variant of the stepper next to the product library for A/B runs (tools/ab_bench.sh):
#   tools/build_variant.sh <name> <extra hipcc flags...>   ->  rlgymppo_cpp_amd/librlgpu_<name>.so  (+ _obj/rlgpu_env_<name>.resource.log / .wwm.log)
# Same steps as the product build (csrc/Makefile): plain hipcc, then tools/wwm_lint.py over the device assembly of that compile as a hard check.
set -e
NAME=$1; shift
cd "$(dirname "$0")/../rlgymppo_cpp_amd/csrc"
make -s all
mkdir -p _obj/tmp_$NAME
/opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC --offload-arch=gfx950 -Wno-unused-value -ffp-contract=off -Rpass-analysis=kernel-resource-usage "$@" -save-temps=obj \
    -c rlgpu_env.hip -o _obj/tmp_$NAME/rlgpu_env.o 2> _obj/rlgpu_env_$NAME.resource.log || { cat _obj/rlgpu_env_$NAME.resource.log; exit 1; }
python3 ../../tools/wwm_lint.py _obj/tmp_$NAME/rlgpu_env-hip-amdgcn-amd-amdhsa-gfx950.s > _obj/rlgpu_env_$NAME.wwm.log || { cat _obj/rlgpu_env_$NAME.wwm.log; exit 1; }
mv _obj/tmp_$NAME/rlgpu_env.o _obj/rlgpu_env_$NAME.o; rm -rf _obj/tmp_$NAME
cat _obj/rlgpu_env_$NAME.wwm.log
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 _obj/rlgpu_env_$NAME.o _obj/rlgpu_learn.o _obj/rlgpu_comm.o _obj/arena_mesh.o _obj/lt_archive.o \
    -o ../librlgpu_$NAME.so -L/opt/rocm/lib -lrccl -lrt -Wl,-rpath,/opt/rocm/lib
grep -A10 "k_env_collectILi2" _obj/rlgpu_env_$NAME.resource.log | grep -E "VGPRs:|AGPRs|Scratch|Spill|LDS Size" | tr -s ' ' | tr '\n' ';'; echo
