#!/bin/bash
# Build an experimental variant of the library for A/B runs (tools/ab_bench.sh, PHNN_LIB_PATH): one translation unit
# is recompiled with extra flags and linked against the product objects.
#   tools/build_variant.sh <name> <mpc|grad|wgrad|split|all> [extra hipcc flags...]   ->  build/ab/lib_<name>.so
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
name="$1"; tu="$2"; shift 2
src="$ROOT/phnn_mpc_amd/csrc"
mkdir -p "$ROOT/build/ab"
make -C "$src" -s >/dev/null
objs=""
for t in mpc grad wgrad split; do
  if [ "$t" = "$tu" ] || [ "$tu" = "all" ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Xclang -target-feature -Xclang -packed-fp32-ops \
      "$@" -c -o "$ROOT/build/ab/${t}_$name.o" "$src/phnn_$t.hip" 2> >(grep -v "is not a recognized feature" >&2) &
    objs="$objs $ROOT/build/ab/${t}_$name.o"
  else
    objs="$objs $src/phnn_$t.o"
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o "$ROOT/build/ab/lib_$name.so" $objs "$src/phnn_pack.o"
rm -f "$ROOT/build/ab/"*_"$name.o"
echo "build/ab/lib_$name.so"
