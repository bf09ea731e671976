export TMPDIR=/tmp
RLGPU_LIB=rlgymppo_cpp_amd/librlgpu_p_inf2.so timeout 300 python3 tools/prof_collect.py 2>&1 | grep -E "per gym step|inference per step|inference share" | tail -6
