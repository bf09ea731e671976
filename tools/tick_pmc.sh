#!/bin/bash
# PMC passes over tools/tick_pmc.py (k_env_ticks only): tools/tick_pmc.sh <tag> [rest|random]   -> gpurun_out/tickpmc_<tag>_<mode>.txt
TAG=$1; MODE=${2:-rest}
OUT=gpurun_out/tickpmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
RES=gpurun_out/tickpmc_${TAG}_${MODE}.txt
: > $RES
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
         "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_BRANCH SQ_INSTS_GDS" \
         "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_IFETCH" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS SQ_INSTS_VALU_CVT SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_IFETCH_LEVEL" \
         "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_INST_REQ"; do
  rocprofv3 --kernel-trace --pmc $C -d $OUT/p$i -o p$i -- python3 tools/tick_pmc.py $MODE 4096 10 > $OUT/p$i.log 2>&1
  f=$(find $OUT/p$i -name '*_results.db' | head -1)
  [ -n "$f" ] && python3 tools/read_prof.py $f | grep -E "k_env_ticks" >> $RES
  tail -1 $OUT/p$i.log >> $RES
  rm -rf $OUT/p$i
  i=$((i+1))
done
cat $RES
