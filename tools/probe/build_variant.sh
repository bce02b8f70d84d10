#!/bin/bash
# usage: build_variant.sh NAME "-DFLAG ..."  -> tools/probe/libcfm_NAME.so (assign.hip rebuilt with the flags; the
# other objects are the product library's).  Measurement infrastructure.
set -e
R="$(cd "$(dirname "$0")/../.." && pwd)"; C="$R/conditional-flow-matching_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -Wno-unused-result $2 -c "$C/assign.hip" -o /tmp/assign_$1.o
objs=""; for f in abi cost sinkhorn sinkhorn_pts transport sample elem mlp mlp_train ode unbalanced; do objs="$objs $C/obj/$f.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/assign_$1.o -o "$R/tools/probe/libcfm_$1.so"
echo "built tools/probe/libcfm_$1.so"
