export TMPDIR=/tmp
for m in tess proc; do echo "== $m"; RLGPU_LIB=rlgymppo_cpp_amd/librlgpu_prof.so timeout 600 python3 tools/prof_collect.py $m 2>&1 | grep -v amdgpu.ids | grep -E "candidate walks" | cut -c1-330; done
VARIANTS="v_ladder v_speed" bash tools/r05ad.sh
