echo "== configs[1], variant = r05 library, tree = inference as a call"; tools/ab_bench.sh rlgymppo_cpp_amd/librlgpu_r05.so 4
echo "== configs[3]"; tools/ab_bench.sh rlgymppo_cpp_amd/librlgpu_r05.so 2 --team-size 2 --envs 8192 --padded-zero-sum --steps 16 --warmup 4
echo "== configs[4]"; tools/ab_bench.sh rlgymppo_cpp_amd/librlgpu_r05.so 2 --team-size 3 --envs 16384 --padded-zero-sum --overlap --fp16 --steps 12 --warmup 3
echo "== lockstep"; tools/ab_bench.sh rlgymppo_cpp_amd/librlgpu_r05.so 2 --steps 40 --warmup 10 --lockstep
