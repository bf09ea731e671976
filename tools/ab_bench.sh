#!/bin/bash
# A/B of stepper builds on one box: tools/ab_bench.sh <variant .so next to librlgpu.so> [rounds]  -- alternating 100-iteration runs of bench_main
V=$1; R=${2:-3}
mkdir -p /tmp/vab && cp rlgymppo_cpp_amd/bench_main rlgymppo_cpp_amd/librlgymppo_amd.so /tmp/vab/ && cp $V /tmp/vab/librlgpu.so
one() { $1 --steps 100 --warmup 10 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2', round(d['value']), round(d['env_kernel_ms_total']/d['env_launches'],3), round(d.get('learn_ms_total',0)/max(d.get('steps',1),1),3))"; }
for i in $(seq $R); do one /tmp/vab/bench_main "variant"; one rlgymppo_cpp_amd/bench_main "tree   "; done
