export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_ref_learner.py tests/test_host_cpp.py -m gpu -x -q -k "composition or deterministic or bench_command" > gpurun_out/r05m_tests.log 2>&1; echo "rc=$?"; tail -15 gpurun_out/r05m_tests.log
for m in "" "--deterministic"; do ./rlgymppo_cpp_amd/bench_main --steps 30 --warmup 8 --lockstep $m 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('lockstep $m value', round(d['value']/1e6,3), 'ppo_iter_ms', d['ppo_iter_ms'])"; done
