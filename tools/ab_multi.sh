#!/bin/bash
# A/B/C... of stepper builds on one box: tools/ab_multi.sh <rounds> <bench_main args in quotes> <name> [<name> ...]
# each <name> = rlgymppo_cpp_amd/librlgpu_<name>.so (tools/build_variant.sh); "tree" = the product library.  Alternating runs of bench_main.
R=$1; ARGS=$2; shift 2
for n in "$@"; do
  mkdir -p /tmp/vab_$n && cp rlgymppo_cpp_amd/bench_main rlgymppo_cpp_amd/librlgymppo_amd.so /tmp/vab_$n/
  if [ "$n" = tree ]; then cp rlgymppo_cpp_amd/librlgpu.so /tmp/vab_$n/librlgpu.so; else cp rlgymppo_cpp_amd/librlgpu_$n.so /tmp/vab_$n/librlgpu.so; fi
done
for i in $(seq $R); do
  for n in "$@"; do
    /tmp/vab_$n/bench_main $ARGS 2>/dev/null | python3 -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print('%-10s' % '$n', round(d['value']), 'collect_ms', round(d['env_kernel_ms_total']/max(d['env_launches'],1),3), 'ppo_ms', round(d.get('ppo_iter_ms',0),3), flush=True)
except Exception as e: print('$n', 'FAILED', e, flush=True)"
  done
done
