export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r05aj_gputests.log 2>&1; echo "gpu tests rc=$?"; grep -E "passed|failed" gpurun_out/r05aj_gputests.log | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --steps 40 --warmup 10 --no-config-legs 2>/dev/null | python3 -c "
import sys, json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in ('value','ms_per_step')}, 'traffic', d['roofline'].get('traffic'), 'frac', d['roofline']['frac'])"
