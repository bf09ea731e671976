#!/bin/bash
# BASELINE configs[3] and configs[4] as worded, on one GPU (bench_main directly; the JSON lines go to gpurun_out/configs_<tag>.txt):
#   [3] 2v2, 8192 envs, padded obs + zero-sum reward          [4] 3v3, 16384 envs, collect-during-learn + fp16 operands with a dynamic loss scale
#   (and [4] without the overlap / with bf16 operands beside it, so each ingredient's cost shows)
TAG=${1:-r04f}; OUT=gpurun_out/configs_$TAG.txt; : > $OUT
EXE=./rlgymppo_cpp_amd/bench_main; export RLGPU_QUIET=1
run() { echo "== $*" >> $OUT; $EXE "$@" 2>/dev/null | tail -1 >> $OUT; }
run --team-size 2 --envs 8192 --padded-zero-sum --steps 40 --warmup 8
run --team-size 3 --envs 16384 --padded-zero-sum --steps 30 --warmup 6
run --team-size 3 --envs 16384 --padded-zero-sum --steps 30 --warmup 6 --fp16
run --team-size 3 --envs 16384 --padded-zero-sum --steps 30 --warmup 6 --overlap
run --team-size 3 --envs 16384 --padded-zero-sum --steps 30 --warmup 6 --overlap --fp16
python3 - $OUT <<'PY'
import sys, json
for line in open(sys.argv[1]):
    if line.startswith("=="): print(line.strip()); continue
    d = json.loads(line); print("   value %.3f M agent-steps/s, %.2f ms per iteration, ppo_iter_ms %.3f, operands %s, collection_during_learn %s, fused_collect %s, %s" % (d["value"] / 1e6, d["ms_per_step"], d["ppo_iter_ms"], d.get("operands"), d.get("collection_during_learn"), d["fused_collect"], d["collection"]))
PY
