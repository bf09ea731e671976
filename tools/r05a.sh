export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05a_gputests.log 2>&1; echo "gpu tests rc=$?" 
tail -3 gpurun_out/r05a_gputests.log
RLGPU_LIB=rlgymppo_cpp_amd/librlgpu_prof.so timeout 600 python3 tools/fine_prof.py 4096 300 > gpurun_out/r05a_fine.txt 2>&1; cat gpurun_out/r05a_fine.txt
timeout 900 tools/tick_pmc.sh r05a rest
timeout 900 tools/tick_pmc.sh r05a random
