export TMPDIR=/tmp RLGPU_QUIET=1
OUT=gpurun_out/prof_r04h_mfma; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES -d $OUT/m -o m -- ./rlgymppo_cpp_amd/bench_main --steps 6 --warmup 3 > $OUT/m.log 2>&1
f=$(find $OUT/m -name '*_results.db' | head -1); python3 tools/read_prof.py $f > $OUT/mfma.summary.txt 2>&1; head -40 $OUT/mfma.summary.txt
find $OUT -name '*.db' -size +20M -delete
