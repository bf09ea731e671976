export TMPDIR=/tmp
MESH=$(python3 -c "import bench; d,i=bench.make_tessellated_mesh_dir(); print(d)")
cd rlgymppo_cpp_amd
cp librlgpu.so librlgpu_new.so
run() { ./bench_main --envs 4096 --steps 20 --warmup 10 $2 2>&1 | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('$1', {k: d.get(k) for k in ('value', 'ms_per_step')}, 'env ms', d['env_kernel_ms_total']/d['env_launches'])"; }
for v in $VARIANTS; do cp librlgpu_$v.so librlgpu.so; run "$v tess" "--mesh-dir $MESH"; run "$v proc" ""; done
cp librlgpu_new.so librlgpu.so
