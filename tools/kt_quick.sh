#!/bin/bash
# kernel trace of a short bench_main run: tools/kt_quick.sh <tag> [bench_main args]
TAG=${1:-q}; shift
ARGS=${@:---steps 4 --warmup 2}
OUT=gpurun_out/kt_$TAG
mkdir -p $OUT
export TMPDIR=/tmp RLGPU_QUIET=1
rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- ./rlgymppo_cpp_amd/bench_main $ARGS > $OUT/kt.log 2>&1
f=$(find $OUT/kt -name '*_results.db' | head -1)
[ -n "$f" ] && python3 tools/read_prof.py $f > $OUT/kt.summary.txt 2>&1
find $OUT -name '*.db' -size +20M -delete
head -14 $OUT/kt.summary.txt
