#!/bin/bash
# usage: build_variant_sk.sh NAME "-DFLAG ..."  -> scratch/variants/NAME.so (sinkhorn.hip rebuilt with the flags)
set -e
R=/root/repo; C=$R/conditional-flow-matching_amd/csrc
mkdir -p $R/scratch/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result $2 -c $C/sinkhorn.hip -o /tmp/sk_$1.o
objs=""; for f in abi cost assign sample elem mlp ode unbalanced; do objs="$objs $C/obj/$f.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/sk_$1.o -o $R/scratch/variants/$1.so
echo built $1
