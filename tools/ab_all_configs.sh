# A/B of two builds of the stepper on the three named configs, interleaved (GPU box):  rlgymppo_cpp_amd/librlgpu_old.so vs the tree's
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for cfg in "1 4096 24" "2 8192 8" "3 16384 6"; do set -- $cfg
  for n in old tree; do
    mkdir -p /tmp/vab_$n && cp rlgymppo_cpp_amd/bench_main rlgymppo_cpp_amd/librlgymppo_amd.so /tmp/vab_$n/
    if [ "$n" = tree ]; then cp rlgymppo_cpp_amd/librlgpu.so /tmp/vab_$n/librlgpu.so; else cp rlgymppo_cpp_amd/librlgpu_$n.so /tmp/vab_$n/librlgpu.so; fi
    extra=""; [ "$1" != 1 ] && extra="--padded-zero-sum"
    /tmp/vab_$n/bench_main --envs $2 --team-size $1 $extra --horizon 32 --steps $3 --warmup 3 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$1v$1 $2 envs %-6s' % '$n', round(d['value']), 'collect_ms', round(d['env_kernel_ms_total']/max(d['env_launches'],1),3), 'ms_per_step', round(d['ms_per_step'],2), flush=True)"
  done
done
done
