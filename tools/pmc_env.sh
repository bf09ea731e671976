#!/bin/bash
# PMC passes over tools/prof_env.py (env stepping only). usage: pmc_env.sh <tag> "<counters pass1>" "<counters pass2>" ...
TAG=$1; shift
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
i=0
for C in "$@"; do
  rocprofv3 --kernel-trace --pmc $C -d $OUT/p$i -o p$i -- python3 tools/prof_env.py 4096 16 > $OUT/p$i.log 2>&1
  f=$(find $OUT/p$i -name '*_results.db' | head -1)
  [ -n "$f" ] && python3 tools/read_prof.py $f | grep -E "k_env_step" 
  rm -rf $OUT/p$i
  i=$((i+1))
done
