#!/bin/bash
# what k_ppo_fwd_bwd costs without its HBM copies / MFMAs: a -DFZ_DEBUG build of the learner in a private copy of the library, kernel traces
cd $GRAFT_REPO_ROOT/rlgymppo_cpp_amd/csrc
mkdir -p /tmp/fzdbg
/opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC --offload-arch=gfx950 -Wno-unused-value -DFZ_DEBUG $FZ_EXTRA -c rlgpu_learn.hip -o /tmp/fzdbg/rlgpu_learn.o || exit 1
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 _obj/rlgpu_env.o /tmp/fzdbg/rlgpu_learn.o _obj/rlgpu_comm.o _obj/arena_mesh.o _obj/lt_archive.o -o /tmp/fzdbg/librlgpu.so -L/opt/rocm/lib -lrccl -lrt -Wl,-rpath,/opt/rocm/lib || exit 1
cp ../bench_main ../librlgymppo_amd.so /tmp/fzdbg/
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp RLGPU_QUIET=1
for dbg in ${FZ_MODES:-0 1 2 3}; do
  OUT=/tmp/fzdbg/kt_$dbg; rm -rf $OUT; mkdir -p $OUT
  RLGPU_FZ_DEBUG=$dbg rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- /tmp/fzdbg/bench_main --steps 4 --warmup 2 > $OUT/kt.log 2>&1
  f=$(find $OUT/kt -name '*_results.db' | head -1)
  echo "debug $dbg: $(python3 tools/read_prof.py $f 2>/dev/null | grep k_ppo_fwd)"
done
