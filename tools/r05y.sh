export TMPDIR=/tmp
bash tools/profile_bench.sh r05y > gpurun_out/r05y_profile.log 2>&1; tail -3 gpurun_out/r05y_profile.log | cut -c1-200
mkdir -p profiles_tmp
( time python bench.py > gpurun_out/r05y_bench.json 2> gpurun_out/r05y_bench.err ) 2>&1 | tail -3; python3 -c "
import json; d=json.load(open('gpurun_out/r05y_bench.json')); print({k:d[k] for k in ('value','ms_per_step','ppo_iter_ms','transport','env_overrides')}); print(d.get('configs')); print('roofline', d.get('roofline')); print('trained_learned', d.get('trained_regime_learned',{}).get('value'), 'tess', d.get('mesh_tessellated',{}).get('value'), 'lockstep', d.get('lockstep_collection',{}).get('value')); print('cpu', d.get('cpu_baseline'))"
