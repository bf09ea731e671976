export TMPDIR=/tmp
for v in p_sub2; do
  echo "== fine_prof $v"; RLGPU_LIB=rlgymppo_cpp_amd/librlgpu_$v.so timeout 600 python3 tools/fine_prof.py 4096 300 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05d_fine_$v.txt
done
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r05d_gputests.log 2>&1; echo "gpu tests rc=$?"; tail -3 gpurun_out/r05d_gputests.log
./rlgymppo_cpp_amd/bench_main --steps 40 --warmup 10 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench_main value', d['value'], 'ms/step', d['ms_per_step'], 'env ms', d['env_kernel_ms_total']/max(1,d['env_launches']), 'ppo', d['ppo_iter_ms'])"
