export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r05at_gputests.log 2>&1; echo "gpu tests rc=$?"; grep -E "passed|failed" gpurun_out/r05at_gputests.log | tail -2
bash tools/profile_bench.sh r05at > gpurun_out/r05at_profile.log 2>&1; tail -3 gpurun_out/r05at_profile.log | cut -c1-200
( time python bench.py > gpurun_out/r05at_bench.json 2> gpurun_out/r05at_bench.err ) 2>&1 | tail -3; python3 -c "
import json; d=json.load(open('gpurun_out/r05at_bench.json')); print({k:d[k] for k in ('value','ms_per_step','ppo_iter_ms','transport','env_overrides')}); print({k:v['value'] for k,v in d.get('configs').items()}); print('traffic', d['roofline'].get('traffic')); print('trained_learned', d.get('trained_regime_learned',{}).get('value'), 'tess', d.get('mesh_tessellated',{}).get('value'), 'lockstep', d.get('lockstep_collection',{}).get('value'))"
