export TMPDIR=/tmp
cd rlgymppo_cpp_amd && for i in 1 2; do ./bench_main --envs 4096 --steps 12 --warmup 3 2>&1 | tail -1 | cut -c1-400; done; cd ..
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r05o_gputests.log 2>&1; echo "gpu tests rc=$?"; tail -5 gpurun_out/r05o_gputests.log
