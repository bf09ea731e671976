export TMPDIR=/tmp
echo "== fine_prof p_sub4"; RLGPU_LIB=rlgymppo_cpp_amd/librlgpu_p_sub4.so timeout 600 python3 tools/fine_prof.py 4096 300 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05g_fine.txt | grep -E "clock|cycles/tick|cand|sum"
RLGPU_LIB=rlgymppo_cpp_amd/librlgpu_p_sub4.so timeout 300 python3 tools/prof_collect.py 2>&1 | tail -12
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r05g_gputests.log 2>&1; echo "gpu tests rc=$?"; tail -3 gpurun_out/r05g_gputests.log
./rlgymppo_cpp_amd/bench_main --steps 40 --warmup 10 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench_main value', d['value'], 'ms/step', d['ms_per_step'], 'env ms', d['env_kernel_ms_total']/max(1,d['env_launches']), 'ppo', d['ppo_iter_ms'])"
