export TMPDIR=/tmp RLGPU_QUIET=1
cd rlgymppo_cpp_amd; cp librlgpu.so librlgpu_new.so; cd ..
for v in v_head new; do
  cp rlgymppo_cpp_amd/librlgpu_$v.so rlgymppo_cpp_amd/librlgpu.so
  OUT=gpurun_out/wr_$v; mkdir -p $OUT
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o write -- ./rlgymppo_cpp_amd/bench_main --steps 6 --warmup 3 > $OUT/write.log 2>&1
  f=$(find $OUT/write -name '*_results.db' | head -1)
  echo "== $v"; python3 tools/read_prof.py $f | grep -E "k_env_collect" | head -3
  rm -rf $OUT/write
done
cp rlgymppo_cpp_amd/librlgpu_new.so rlgymppo_cpp_amd/librlgpu.so
