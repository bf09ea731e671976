#!/bin/bash
# rocprofv3 passes over bench.py (run on the GPU box via gpurun): kernel trace + two PMC passes; text summaries in gpurun_out/prof_<tag>/
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python3 bench.py --steps 3 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/kt.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/write.log 2>&1
for p in kt fetch write; do
  f=$(find $OUT/$p -name '*_results.db' | head -1)
  [ -n "$f" ] && python3 tools/read_prof.py $f > $OUT/$p.summary.txt 2>&1
done
find $OUT -name '*.db' -size +20M -delete
cat $OUT/bench.json; head -30 $OUT/kt.summary.txt; grep -i "k_env_step\|k_gemm" $OUT/fetch.summary.txt $OUT/write.summary.txt | head
