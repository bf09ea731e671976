#!/bin/bash
# A/B of stepper builds on one box: tools/ab_bench.sh <variant .so next to librlgpu.so> [rounds] [bench_main args...]  -- alternating runs of bench_main
# (default 100 iterations of BASELINE configs[1]); prints value, collection launch ms, PPO ms per iteration for each run
V=$1; R=${2:-3}; shift; shift
ARGS=${@:---steps 100 --warmup 10}
mkdir -p /tmp/vab && cp rlgymppo_cpp_amd/bench_main rlgymppo_cpp_amd/librlgymppo_amd.so /tmp/vab/ && cp $V /tmp/vab/librlgpu.so
one() { $1 $ARGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2', round(d['value']), round(d['env_kernel_ms_total']/max(d['env_launches'],1),3), round(d.get('ppo_iter_ms',0),3))"; }
for i in $(seq $R); do one /tmp/vab/bench_main "variant"; one rlgymppo_cpp_amd/bench_main "tree   "; done
