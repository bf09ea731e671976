export TMPDIR=/tmp
timeout 1200 python3 tools/soak.py 0 400 2 4096 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-900
