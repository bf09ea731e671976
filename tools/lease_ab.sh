echo "== configs[1], variant = uniform tick arguments (NC <= 4), tree = vector arguments"; tools/ab_bench.sh rlgymppo_cpp_amd/librlgpu_uni4.so 4
echo "== configs[3]"; tools/ab_bench.sh rlgymppo_cpp_amd/librlgpu_uni4.so 2 --team-size 2 --envs 8192 --padded-zero-sum --steps 16 --warmup 4
echo "== plugin_fallback_check"; (cd rlgymppo_cpp_amd && timeout 600 ./plugin_fallback_check) 2>&1 | tail -25
echo "== user reward"; for i in 1 2; do ./rlgymppo_cpp_amd/bench_main --steps 60 --warmup 10 --user-reward 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('deferred', round(d['value']), round(d['ms_per_step'],2), round(d['ppo_iter_ms'],2), d.get('user_reward'), d['fused_collect'], d['collection'])"; done
./rlgymppo_cpp_amd/bench_main --steps 8 --warmup 2 --user-reward --no-defer 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('per-step', round(d['value']), round(d['ms_per_step'],2), d.get('user_reward'))"
