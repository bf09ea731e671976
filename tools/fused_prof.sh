#!/bin/bash
# per-phase cycles of k_ppo_fwd_bwd: builds the learner with -DFZ_PROF into a private copy of the library on the box and runs bench_main with it
cd $GRAFT_REPO_ROOT/rlgymppo_cpp_amd/csrc
mkdir -p /tmp/fzprof
/opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC --offload-arch=gfx950 -Wno-unused-value -DFZ_PROF -c rlgpu_learn.hip -o /tmp/fzprof/rlgpu_learn.o || exit 1
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 _obj/rlgpu_env.o /tmp/fzprof/rlgpu_learn.o _obj/rlgpu_comm.o _obj/arena_mesh.o _obj/lt_archive.o -o /tmp/fzprof/librlgpu.so -L/opt/rocm/lib -lrccl -lrt -Wl,-rpath,/opt/rocm/lib || exit 1
cp ../bench_main ../librlgymppo_amd.so /tmp/fzprof/
RLGPU_QUIET=1 RLGPU_FUSED_PROF=1 /tmp/fzprof/bench_main --steps 8 --warmup 2 "$@" 2>&1 | grep "k_ppo_fwd" | tail -2
