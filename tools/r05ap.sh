export TMPDIR=/tmp
RLGPU_LIB=rlgymppo_cpp_amd/librlgpu_prof.so timeout 600 python3 tools/prof_teams.py 2 8192 16 2>&1 | grep -v amdgpu.ids | tail -4
RLGPU_LIB=rlgymppo_cpp_amd/librlgpu_prof.so timeout 600 python3 tools/prof_teams.py 3 8190 12 2>&1 | grep -v amdgpu.ids | tail -4
