export TMPDIR=/tmp
cd rlgymppo_cpp_amd
cp librlgpu.so librlgpu_new.so
run() { ./bench_main --envs 4096 --steps 30 --warmup 10 $2 2>&1 | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('$1', d['value'], 'env ms', d['env_kernel_ms_total']/d['env_launches'])"; }
for rep in 1 2 3; do for v in new infcall; do cp librlgpu_$v.so librlgpu.so; run "$v"; done; done
for v in new infcall; do cp librlgpu_$v.so librlgpu.so; run "$v 3v3" "--team-size 3 --envs 4096 --padded-zero-sum --steps 8 --warmup 3"; done
cp librlgpu_new.so librlgpu.so
