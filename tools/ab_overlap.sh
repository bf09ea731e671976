EXE=./rlgymppo_cpp_amd/bench_main; export RLGPU_QUIET=1
for rep in 1 2 3; do for v in "" "--overlap" "--fp16" "--overlap --fp16"; do
  echo -n "rep $rep [$v] "; $EXE --team-size 3 --envs 16384 --padded-zero-sum --steps 20 --warmup 4 $v 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f M  %.2f ms' % (d['value']/1e6, d['ms_per_step']))"
done; done
