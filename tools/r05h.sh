export TMPDIR=/tmp
RLGPU_LIB=rlgymppo_cpp_amd/librlgpu_p_inf4.so timeout 300 python3 tools/prof_collect.py 2>&1 | grep -E "per gym step|inference per step" | tail -2
cp rlgymppo_cpp_amd/librlgpu.so /tmp/new.so; cp rlgymppo_cpp_amd/librlgpu_v_prev.so /tmp/old.so
for rep in 1 2; do for v in old new; do
  cp /tmp/$v.so rlgymppo_cpp_amd/librlgpu.so
  ./rlgymppo_cpp_amd/bench_main --steps 40 --warmup 10 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v bench_main value', d['value'], 'ms/step', d['ms_per_step'], 'env ms', d['env_kernel_ms_total']/max(1,d['env_launches']), 'ppo', d['ppo_iter_ms'])"
done; done
cp /tmp/new.so rlgymppo_cpp_amd/librlgpu.so
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r05h_gputests.log 2>&1; echo "gpu tests rc=$?"; tail -3 gpurun_out/r05h_gputests.log
