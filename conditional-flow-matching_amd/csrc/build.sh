#!/bin/bash
# Build libcfm_gfx950.so (gfx950 only) next to this script's package directory.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/../libcfm_gfx950.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -Wno-unused-result $CFM_EXTRA_FLAGS"
mkdir -p "$HERE/obj"
objs=""
for f in abi cost sinkhorn sinkhorn_pts assign transport sample elem mlp mlp_train ode unbalanced; do
  src="$HERE/$f.hip"; obj="$HERE/obj/$f.o"
  stale=0
  if [ ! -f "$obj" ] || [ "$src" -nt "$obj" ]; then stale=1; fi
  for h in "$HERE"/*.h "$HERE"/../../include/*.h; do      # every header of the library: a stale object is worse than a rebuild
    if [ "$h" -nt "$obj" ]; then stale=1; fi
  done
  if [ $stale = 1 ]; then
    "$HIPCC" $FLAGS -c "$src" -o "$obj" &
  fi
  objs="$objs $obj"
done
wait
"$HIPCC" --offload-arch=gfx950 -shared -fPIC $objs -o "$OUT"
echo "built $OUT"
