export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r05k_gputests.log 2>&1; echo "gpu tests rc=$?"; tail -5 gpurun_out/r05k_gputests.log
( time python bench.py > gpurun_out/r05k_bench.json 2> gpurun_out/r05k_bench.err ) 2>&1 | tail -3; python3 -c "
import json; d=json.load(open('gpurun_out/r05k_bench.json')); print({k:d[k] for k in ('value','ms_per_step','ppo_iter_ms','transport','env_overrides')}); print(d.get('configs')); print('trained_learned', d.get('trained_regime_learned',{}).get('value'), 'tess', d.get('mesh_tessellated',{}).get('value'), 'lockstep', d.get('lockstep_collection',{}).get('value'))"
