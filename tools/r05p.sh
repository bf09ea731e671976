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
cp librlgpu_new.so librlgpu.so
HSA_SCRATCH_SINGLE_LIMIT=4000000000 run new_scratchlimit
# the plugin check with the skill tracker case
cd .. && mkdir -p /tmp/pf && cd /tmp/pf && timeout 600 $GRAFT_REPO_ROOT/rlgymppo_cpp_amd/plugin_fallback_check 2>&1 | tail -12
