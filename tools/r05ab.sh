export TMPDIR=/tmp
RLGPU_LIB=rlgymppo_cpp_amd/librlgpu_prof.so timeout 900 python3 tools/fine_prof.py 4096 300 0 tess 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05ab_fine_tess.txt | tail -45
