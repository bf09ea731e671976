export TMPDIR=/tmp
cd rlgymppo_cpp_amd
run() { ./bench_main $2 2>&1 | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('$1', d['value'], 'ms/iter', d['ms_per_step'], 'env ms', d['env_kernel_ms_total']/max(1,d['env_launches']))"; }
for q in 0 1; do
run "c1 lockstep queue=$q" "--envs 4096 --lockstep --steps 20 --warmup 5 --collect-queue $q"
run "2v2 2048 lockstep queue=$q" "--team-size 2 --envs 2048 --lockstep --steps 20 --warmup 5 --collect-queue $q"
done
run "c1 free" "--envs 4096 --steps 20 --warmup 5"
