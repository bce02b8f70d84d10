#!/bin/bash
# usage: build_variant_all.sh NAME "-flags ..." [files...]  -> tools/probe/libcfm_NAME.so
# The listed sources (default: cost mlp mlp_train ode) are rebuilt with the extra flags, the other objects are the
# product library's.  Load with CFM_LIB_PATH=tools/probe/libcfm_NAME.so (an A/B measurement switch, never a fallback).
set -e
R="$(cd "$(dirname "$0")/../.." && pwd)"; C="$R/conditional-flow-matching_amd/csrc"
NAME=$1; FL=$2; shift; shift
FILES="${*:-cost mlp mlp_train ode}"
objs=""
for f in abi cost sinkhorn sinkhorn_pts assign transport sample elem mlp mlp_train ode unbalanced; do
  if [[ " $FILES " == *" $f "* ]]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -Wno-unused-result $FL -c "$C/$f.hip" -o /tmp/${f}_$NAME.o &
    objs="$objs /tmp/${f}_$NAME.o"
  else
    objs="$objs $C/obj/$f.o"
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o "$R/tools/probe/libcfm_$NAME.so"
echo "built tools/probe/libcfm_$NAME.so"
