export TMPDIR=/tmp
cd rlgymppo_cpp_amd
cp librlgpu.so librlgpu_new.so
cp librlgpu_v_wwmfast.so librlgpu.so
./bench_main --envs 4096 --steps 6 --warmup 2 2>&1 | tail -5 | cut -c1-400
cp librlgpu_new.so librlgpu.so
