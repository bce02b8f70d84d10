#!/bin/bash
# Build libcfm_gfx950.so (gfx950 only) next to this script's package directory.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/../libcfm_gfx950.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -Wno-unused-result $CFM_EXTRA_FLAGS"
mkdir -p "$HERE/obj"
objs=""; pids=""
for f in abi cost sinkhorn sinkhorn_pts assign transport sample elem mlp mlp_train ode unbalanced; do
  src="$HERE/$f.hip"; obj="$HERE/obj/$f.o"
  stale=0
  if [ ! -f "$obj" ] || [ "$src" -nt "$obj" ]; then stale=1; fi
  for h in "$HERE"/*.h "$HERE"/../../include/*.h; do      # every header of the library: a stale object is worse than a rebuild
    if [ "$h" -nt "$obj" ]; then stale=1; fi
  done
  if [ $stale = 1 ]; then
    rm -f "$obj"                       # a failed compile must not leave the previous object to be linked
    "$HIPCC" $FLAGS -c "$src" -o "$obj" &
    pids="$pids $!"
  fi
  objs="$objs $obj"
done
fail=0
for p in $pids; do wait $p || fail=1; done      # (a bare `wait` returns 0 whatever the compilers said)
if [ $fail = 1 ]; then echo "build.sh: a translation unit failed to compile" >&2; exit 1; fi
"$HIPCC" --offload-arch=gfx950 -shared -fPIC $objs -o "$OUT"
echo "built $OUT"
