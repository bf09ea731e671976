#!/bin/bash
# quick experiment variant of the stepper (prof or plain), NC=2 only: /tmp/bv.sh <name> <flags...>
NAME=$1; shift
cd /root/repo/rlgymppo_cpp_amd/csrc
HIPCC=/opt/rocm/bin/hipcc python3 ../../tools/hipcc_wwm_safe.py --log _obj/rlgpu_env_$NAME.wwm.log -std=c++17 -O3 -fPIC --offload-arch=gfx950 -Wno-unused-value -ffp-contract=off -DRLG_ONLY_NC2 -Rpass-analysis=kernel-resource-usage "$@" \
    -c rlgpu_env.hip -o _obj/rlgpu_env_$NAME.o 2> _obj/rlgpu_env_$NAME.resource.log || { tail -20 _obj/rlgpu_env_$NAME.resource.log; exit 1; }
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 _obj/rlgpu_env_$NAME.o _obj/rlgpu_learn.o _obj/rlgpu_comm.o _obj/arena_mesh.o _obj/lt_archive.o \
    -o ../librlgpu_$NAME.so -L/opt/rocm/lib -lrccl -lrt -Wl,-rpath,/opt/rocm/lib
echo -n "$NAME: "; grep -A14 "k_env_ticksILi2" _obj/rlgpu_env_$NAME.resource.log | grep -E "VGPRs:|AGPRs|Scratch|SGPRs Spill|VGPRs Spill" | sed 's/.*remark: *//; s/ \[-Rpass.*//' | tr '\n' ';'; echo
