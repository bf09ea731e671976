export TMPDIR=/tmp
RLGPU_LIB=rlgymppo_cpp_amd/librlgpu_prof.so timeout 300 python3 tools/prof_collect.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05t_prof_collect.txt | tail -30
