cd $GRAFT_REPO_ROOT
for i in 1 2 3; do for n in tree tn64 tn128; do
  mkdir -p /tmp/vab_$n && cp rlgymppo_cpp_amd/bench_main rlgymppo_cpp_amd/librlgymppo_amd.so /tmp/vab_$n/
  if [ "$n" = tree ]; then cp rlgymppo_cpp_amd/librlgpu.so /tmp/vab_$n/librlgpu.so; else cp rlgymppo_cpp_amd/librlgpu_$n.so /tmp/vab_$n/librlgpu.so; fi
  /tmp/vab_$n/bench_main --envs 4096 --team-size 1 --horizon 32 --steps 40 --warmup 10 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%-8s' % '$n', 'ppo_ms', round(d.get('ppo_iter_ms',0),3), 'gemm ms/minibatch', round(d['gemm_ms_total']/max(d['gemm_calls'],1),4), 'TFLOP/s', round(d['gemm_flops_total']/max(d['gemm_ms_total'],1e-9)/1e9,1), flush=True)"
done; done
