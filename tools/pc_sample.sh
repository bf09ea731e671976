#!/bin/bash
# PC sampling of the collection kernel on the GPU box (rocprofv3 beta feature): tools/pc_sample.sh <tag> [bench_main args]
TAG=${1:-pcs}; shift
ARGS=${@:---steps 4 --warmup 2}
OUT=gpurun_out/pcs_$TAG
mkdir -p $OUT
export TMPDIR=/tmp RLGPU_QUIET=1 ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
EXE=./rlgymppo_cpp_amd/bench_main
rocprofv3 -L > $OUT/avail.txt 2>&1
timeout 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method stochastic --pc-sampling-unit cycles --pc-sampling-interval 1048576 --output-format csv json -d $OUT/st -o st -- $EXE $ARGS > $OUT/st.log 2>&1
echo "stochastic rc=$?"
if ! find $OUT/st -type f | grep -q .; then
  timeout 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method host_trap --pc-sampling-unit time --pc-sampling-interval 100 --output-format csv json -d $OUT/ht -o ht -- $EXE $ARGS > $OUT/ht.log 2>&1
  echo "host_trap rc=$?"
fi
find $OUT -type f | xargs ls -la
tail -20 $OUT/st.log
