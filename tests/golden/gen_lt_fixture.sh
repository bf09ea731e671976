#!/bin/sh
# Regenerates the .lt fixtures with the libtorch that ships inside the torch wheel (CPU).  Usage: sh tests/golden/gen_lt_fixture.sh
set -e
here=$(cd "$(dirname "$0")" && pwd)
TI=$(python3 -c "import torch,os;print(os.path.dirname(torch.__file__))")
g++ -std=c++17 -O0 "$here/gen_lt_fixture.cpp" -I"$TI/include" -I"$TI/include/torch/csrc/api/include" -L"$TI/lib" -ltorch -ltorch_cpu -lc10 -Wl,-rpath,"$TI/lib" -o /tmp/gen_lt_fixture
/tmp/gen_lt_fixture "$here"
