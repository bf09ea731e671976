#!/bin/bash
# A/B of two complete builds: $1 = directory with bench_main + its libraries (variant), the tree's own = "tree"
V=$1; R=${2:-4}; shift; shift
ARGS=${@:---steps 200 --warmup 20}
one() { $1 $ARGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2', round(d['value']), round(d['env_kernel_ms_total']/max(d['env_launches'],1),3), round(d.get('ppo_iter_ms',0),3))"; }
for i in $(seq $R); do one $V/bench_main "base   "; one rlgymppo_cpp_amd/bench_main "tree   "; done
