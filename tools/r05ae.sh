export TMPDIR=/tmp
cd rlgymppo_cpp_amd
cp librlgpu.so librlgpu_new.so
run() { ./bench_main --envs 4096 --steps 30 --warmup 10 2>&1 | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('$1', d['value'], 'env ms', d['env_kernel_ms_total']/d['env_launches'])"; }
for rep in 1 2 3 4; do for v in $VARIANTS; do cp librlgpu_$v.so librlgpu.so; run "$v"; done; done
cp librlgpu_new.so librlgpu.so
