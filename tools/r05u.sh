export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r05u_gputests.log 2>&1; echo "gpu tests rc=$?"; grep -E "passed|failed|Error" gpurun_out/r05u_gputests.log | tail -5
