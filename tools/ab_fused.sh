#!/bin/bash
# fused minibatch path vs per-layer path: bench_main numbers on one box, alternating
export RLGPU_QUIET=1
for i in 1 2; do for mode in fused perlayer; do
  if [ $mode = perlayer ]; then export RLGPU_NO_FUSED=1; else unset RLGPU_NO_FUSED; fi
  ./rlgymppo_cpp_amd/bench_main --envs 4096 --team-size 1 --horizon 32 --steps 40 --warmup 10 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%-9s' % '$mode', 'value', round(d['value']/1e6,2), 'ms/iter', round(d['ms_per_step'],2), 'ppo_ms', round(d.get('ppo_iter_ms',0),3), 'gemm ms/minibatch', round(d['gemm_ms_total']/max(d['gemm_calls'],1),4), 'TFLOP/s', round(d['gemm_flops_total']/max(d['gemm_ms_total'],1e-9)/1e9,1), flush=True)"
done; done
