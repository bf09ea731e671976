cd /tmp && export TMPDIR=/tmp RLGPU_QUIET=1; cd $GRAFT_REPO_ROOT
for n in "$@"; do
  mkdir -p /tmp/vab_$n && cp rlgymppo_cpp_amd/bench_main rlgymppo_cpp_amd/librlgymppo_amd.so /tmp/vab_$n/
  if [ "$n" = tree ]; then cp rlgymppo_cpp_amd/librlgpu.so /tmp/vab_$n/librlgpu.so; else cp rlgymppo_cpp_amd/librlgpu_$n.so /tmp/vab_$n/librlgpu.so; fi
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/ws_$n -o w -- /tmp/vab_$n/bench_main --steps 6 --warmup 3 > /tmp/ws_$n.log 2>&1
  f=$(find /tmp/ws_$n -name '*_results.db' | head -1); echo "== $n"; python3 tools/read_prof.py $f 2>&1 | grep -i "collect" | head -4
done
