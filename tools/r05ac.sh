export TMPDIR=/tmp
RLGPU_LIB=rlgymppo_cpp_amd/librlgpu_prof.so timeout 600 python3 tools/prof_collect.py tess 2>&1 | grep -v amdgpu.ids | grep -E "per gym step|inference per step|workgroup cycles min" | tail -6
