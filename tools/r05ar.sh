export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "step_queue" 2>&1 | grep -E "^E|Error|passed|failed" | head -12
