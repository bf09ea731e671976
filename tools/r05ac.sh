export TMPDIR=/tmp
for m in tess proc; do echo "== $m"; RLGPU_LIB=rlgymppo_cpp_amd/librlgpu_prof.so timeout 600 python3 tools/prof_collect.py $m 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-300; done
