export TMPDIR=/tmp
cd rlgymppo_cpp_amd
cp librlgpu.so librlgpu_new.so
run() { ./bench_main --envs 4096 --steps 12 --warmup 3 2>&1 | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('$1', {k: d.get(k) for k in ('value', 'ms_per_step', 'env_kernel_ms_total', 'ppo_iter_ms')})"; }
for rep in 1 2; do
  cp librlgpu_v_head.so librlgpu.so; run head
  cp librlgpu_new.so librlgpu.so; run new
done
cp librlgpu_new.so librlgpu.so; cd ..
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r05s_gputests.log 2>&1; echo "gpu tests rc=$?"; grep -E "passed|failed" gpurun_out/r05s_gputests.log | tail -3
