tools/profile_bench.sh r04h > gpurun_out/prof_r04h.out 2>&1
cp gpurun_out/prof_r04h/r04h_pmc.json profiles/r04h_pmc.json
python bench.py > gpurun_out/bench_r04h.json 2> gpurun_out/bench_r04h.err
tail -c 600 gpurun_out/bench_r04h.json
for s in 1024 4096; do echo "slab $s"; RLGPU_DW_SLAB=$s ./rlgymppo_cpp_amd/bench_main --steps 30 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d.get('ppo_iter_ms'))"; done
tools/configs_legs.sh r04h
