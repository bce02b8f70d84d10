#!/bin/bash
# tools/probe/libcfm_prof.so = the product library with assign.hip rebuilt under -DSP_PROFILE (cycle counters of the
# one-workgroup list solver, read by tools/asg_solver_prof.py).  Measurement infrastructure only.
set -e
R="$(cd "$(dirname "$0")/../.." && pwd)"; C="$R/conditional-flow-matching_amd/csrc"
bash "$C/build.sh" > /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -Wno-unused-result -DSP_PROFILE $1 -c "$C/assign.hip" -o /tmp/assign_prof.o
objs=""; for f in abi cost sinkhorn sinkhorn_pts transport sample elem mlp mlp_train ode unbalanced; do objs="$objs $C/obj/$f.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/assign_prof.o -o "$R/tools/probe/libcfm_prof.so"
echo "built tools/probe/libcfm_prof.so"
