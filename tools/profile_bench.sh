#!/bin/bash
# rocprofv3 passes over the measured process of bench.py (rlgymppo_cpp_amd/bench_main, run DIRECTLY under the profiler -- no Python,
# no launcher hop), on the GPU box via gpurun:  tools/profile_bench.sh <tag> [bench_main args]
#   kernel trace + stats | FETCH_SIZE | WRITE_SIZE | two SQ passes  ->  gpurun_out/prof_<tag>/{kt,fetch,write,sq1,sq2}.summary.txt + <tag>_pmc.json
TAG=${1:-r02}; shift
ARGS=${@:---steps 6 --warmup 3}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp RLGPU_QUIET=1
EXE=./rlgymppo_cpp_amd/bench_main
$EXE $ARGS > $OUT/bench_main.json 2> $OUT/bench_main.err
rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- $EXE $ARGS > $OUT/kt.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o fetch -- $EXE $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o write -- $EXE $ARGS > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d $OUT/sq1 -o sq1 -- $EXE $ARGS > $OUT/sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE -d $OUT/sq2 -o sq2 -- $EXE $ARGS > $OUT/sq2.log 2>&1
for p in kt fetch write sq1 sq2; do
  f=$(find $OUT/$p -name '*_results.db' | head -1)
  [ -n "$f" ] && python3 tools/read_prof.py $f > $OUT/$p.summary.txt 2>&1
done
python3 tools/pmc_summary.py $OUT $TAG > $OUT/${TAG}_pmc.json 2> $OUT/pmc_summary.err
find $OUT -name '*.db' -size +20M -delete
cat $OUT/bench_main.json; head -12 $OUT/kt.summary.txt; cat $OUT/${TAG}_pmc.json
