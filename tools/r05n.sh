export TMPDIR=/tmp
timeout 1400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "no_contact_is_ever_dropped or no_kernel_writes_past or wedge" > gpurun_out/r05n_tiny.log 2>&1; echo "tiny test rc=$?"; grep -E "cut-down|wedge fixture|passed|failed|Error|assert" gpurun_out/r05n_tiny.log | tail -12
cd rlgymppo_cpp_amd && for i in 1 2; do ./bench_main --envs 4096 --steps 12 --warmup 3 2>&1 | tail -2 | cut -c1-600; done
