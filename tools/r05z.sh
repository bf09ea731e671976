export TMPDIR=/tmp
timeout 1500 python3 tools/soak.py 600 0 2>&1 | grep -v amdgpu.ids | tail -2 > gpurun_out/r05z_soak.txt
timeout 1500 python3 tools/soak.py 0 4000 3 1024 2>&1 | grep -v amdgpu.ids | tail -1 >> gpurun_out/r05z_soak.txt
timeout 900 python3 tools/soak.py 0 1500 2 1024 2>&1 | grep -v amdgpu.ids | tail -1 >> gpurun_out/r05z_soak.txt
timeout 900 python3 tools/soak.py 0 2000 1 2048 2>&1 | grep -v amdgpu.ids | tail -1 >> gpurun_out/r05z_soak.txt
cat gpurun_out/r05z_soak.txt | cut -c1-400
