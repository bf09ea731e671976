export TMPDIR=/tmp
for a in "1 256 0 1" "1 1024 0 1" "2 256 0 1"; do timeout 300 python3 tools/determinism/dbg_state_diff.py $a 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-200; done
for a in "1 512 1 3 free" "1 1024 1 3 lock" "3 171 1 3 free"; do timeout 300 python3 tools/determinism/dbg_free.py $a 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-200; done
