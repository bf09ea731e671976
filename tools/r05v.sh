export TMPDIR=/tmp
cd rlgymppo_cpp_amd
cp librlgpu.so librlgpu_new.so
run() { ./bench_main --envs 4096 --steps 12 --warmup 3 2>&1 | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('$1', {k: d.get(k) for k in ('value', 'ms_per_step', 'env_kernel_ms_total', 'ppo_iter_ms')})"; }
for rep in 1 2; do
  for v in $VARIANTS; do cp librlgpu_$v.so librlgpu.so; run $v; done
done
cp librlgpu_new.so librlgpu.so
