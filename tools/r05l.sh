export TMPDIR=/tmp
cp rlgymppo_cpp_amd/librlgpu.so /tmp/keep.so
MESH=$(python3 -c "import bench; d,i=bench.make_tessellated_mesh_dir(); print(d)")
for rep in 1 2; do for v in v_r04head v_exact v_cube; do
  cp rlgymppo_cpp_amd/librlgpu_$v.so rlgymppo_cpp_amd/librlgpu.so
  for m in proc tess; do
    if [ $m = tess ]; then A="--mesh-dir $MESH"; else A=""; fi
    ./rlgymppo_cpp_amd/bench_main --steps 40 --warmup 10 $A 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v $m value', round(d['value']/1e6,3), 'env ms', round(d['env_kernel_ms_total']/max(1,d['env_launches']),3))"
  done
done; done
cp /tmp/keep.so rlgymppo_cpp_amd/librlgpu.so
