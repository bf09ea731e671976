#!/bin/bash
mkdir -p /tmp/rdvx && chmod 700 /tmp/rdvx
export WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=29577 LOCAL_RANK=0 RLGPU_COMM_DIR=/tmp/rdvx RLGPU_COMM_TAG=x RLGPU_COMM_TRANSPORT=shm RLGPU_COMM_TIMEOUT_S=20 RLGPU_REPLICA_CHECK_EVERY=1 RLGPU_LOCKSTEP_COLLECTION=1 RLGPU_QUIET=1
RANK=0 ./rlgymppo_cpp_amd/bench_main --envs 256 --horizon 8 --steps 3000 --warmup 0 > /tmp/r0.out 2> /tmp/r0.err &
RANK=1 ./rlgymppo_cpp_amd/bench_main --envs 256 --horizon 8 --steps 3000 --warmup 0 > /tmp/r1.out 2> /tmp/r1.err
wait
echo "--- r0"; tail -5 /tmp/r0.err; echo "--- r1"; tail -5 /tmp/r1.err
