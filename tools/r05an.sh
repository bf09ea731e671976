export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r05an_gputests.log 2>&1; echo "gpu tests rc=$?"; grep -E "passed|failed" gpurun_out/r05an_gputests.log | tail -2
