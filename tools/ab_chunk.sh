#!/bin/bash
export RLGPU_QUIET=1
for cfg in "0 2048" "16384 512" "16384 1024" "16384 2048" "32768 2048" "32768 1024" "8192 1024"; do
  set -- $cfg
  RLGPU_FUSED_CHUNK=$1 RLGPU_DW_SLAB=$2 ./rlgymppo_cpp_amd/bench_main --envs 4096 --team-size 1 --horizon 32 --steps 30 --warmup 8 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('chunk %6s slab %5s' % ('$1','$2'), 'ppo_ms', round(d.get('ppo_iter_ms',0),3), 'gemm ms/minibatch', round(d['gemm_ms_total']/max(d['gemm_calls'],1),4), 'TFLOP/s', round(d['gemm_flops_total']/max(d['gemm_ms_total'],1e-9)/1e9,1), flush=True)"
done
