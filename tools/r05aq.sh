export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "step_queue or fused_collection_equals" 2>&1 | tail -4
cd rlgymppo_cpp_amd
run() { ./bench_main $2 2>&1 | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('$1', d['value'], 'ms/iter', d['ms_per_step'], 'env ms', d['env_kernel_ms_total']/max(1,d['env_launches']))"; }
for q in 0 -1; do
run "c3 queue=$q" "--team-size 2 --envs 8192 --padded-zero-sum --steps 16 --warmup 4 --collect-queue $q"
run "c4 queue=$q" "--team-size 3 --envs 16384 --padded-zero-sum --overlap --fp16 --steps 12 --warmup 3 --collect-queue $q"
run "c1 lockstep queue=$q" "--envs 4096 --lockstep --steps 20 --warmup 5 --collect-queue $q"
done
