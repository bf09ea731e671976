export TMPDIR=/tmp
cp rlgymppo_cpp_amd/librlgpu.so /tmp/keep.so
for rep in 1 2; do for v in v_r04head v_f2 v_f3 v_cube2 v_cube3; do
  cp rlgymppo_cpp_amd/librlgpu_$v.so rlgymppo_cpp_amd/librlgpu.so
  ./rlgymppo_cpp_amd/bench_main --steps 40 --warmup 10 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v bench_main value', d['value'], 'ms/step', d['ms_per_step'], 'env ms', d['env_kernel_ms_total']/max(1,d['env_launches']), 'ppo', d['ppo_iter_ms'])"
done; done
cp /tmp/keep.so rlgymppo_cpp_amd/librlgpu.so
