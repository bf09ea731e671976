export TMPDIR=/tmp
RLGPU_LIB=rlgymppo_cpp_amd/librlgpu_prof.so timeout 1500 python3 tools/fine_prof.py 4096 300 600 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05ao_fine_learned.txt | tail -45
