cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
for v in "" _epalds; do
  if [ -z "$v" ]; then unset RLGPU_LIB; else export RLGPU_LIB=rlgymppo_cpp_amd/librlgpu$v.so; fi
  python tools/epa_probe.py 2>&1 | awk -v v="tree$v" '/launch/ {n++; if (n>2) {s+=$3; q+=$7}} END {printf "%s mean ms %.3f  EPA queries/launch %.0f\n", v, s/(n-2), q/(n-2)}'
done; done
