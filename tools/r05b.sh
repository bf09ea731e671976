export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r05b_gputests.log 2>&1; echo "gpu tests rc=$?"; tail -3 gpurun_out/r05b_gputests.log
for v in p_base p_noslp; do
  echo "== fine_prof $v"; RLGPU_LIB=rlgymppo_cpp_amd/librlgpu_$v.so timeout 600 python3 tools/fine_prof.py 4096 300 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05b_fine_$v.txt
done
for rep in 1 2; do for v in tree v_noslp; do
  if [ $v = tree ]; then unset RLGPU_LIB; else export RLGPU_LIB=rlgymppo_cpp_amd/librlgpu_$v.so; fi
  echo "== ticks $v"; timeout 300 python3 tools/tick_pmc.py rest 4096 20 2>&1 | tail -1; timeout 300 python3 tools/tick_pmc.py random 4096 20 300 2>&1 | tail -1
done; done
