export TMPDIR=/tmp
RLGPU_LIB=rlgymppo_cpp_amd/librlgpu_prof.so timeout 600 python3 tools/fine_prof.py 4096 300 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05r_fine.txt | tail -60
