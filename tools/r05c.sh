export TMPDIR=/tmp
for v in p_sub; do
  echo "== fine_prof $v"; RLGPU_LIB=rlgymppo_cpp_amd/librlgpu_$v.so timeout 600 python3 tools/fine_prof.py 4096 300 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05c_fine_$v.txt
done
