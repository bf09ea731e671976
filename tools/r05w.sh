export TMPDIR=/tmp
bash tools/profile_bench.sh r05w 2>&1 | tail -25
cp gpurun_out/prof_r05w/bench_py.json gpurun_out/r05w_bench_py.json 2>/dev/null
