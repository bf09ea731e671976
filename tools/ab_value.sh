#!/bin/bash
export RLGPU_QUIET=1
for i in 1 2; do for mode in stripe infer; do
  if [ $mode = infer ]; then export RLGPU_NO_VALUE_STRIPE=1; else unset RLGPU_NO_VALUE_STRIPE; fi
  ./rlgymppo_cpp_amd/bench_main --envs 4096 --team-size 1 --horizon 32 --steps 40 --warmup 10 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%-7s' % '$mode', 'value', round(d['value']/1e6,2), 'ms/iter', round(d['ms_per_step'],2), 'ppo_ms', round(d.get('ppo_iter_ms',0),3), flush=True)"
done; done
