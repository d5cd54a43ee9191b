#!/bin/bash
# Lab build of the library with one source recompiled under extra flags:
#   scripts/lab/build_variant.sh [-H] <name> <file.hip> "<flags>"   ->  scripts/lab/_build/libacr_<name>.so  (use with ACR_LIB_PATH)
# -H: take <file.hip> (and its headers) from the last commit that carried the lab timing hooks (-DLAB_TL / -DLAB_STAMP / -DLAB_TLB /
#     -DSPLIT_SPREAD phase timers and stamps; round 6 stripped them from the product sources, scripts/lab/strip_lab_hooks.py --
#     the device ISA of the product build did not change).  The hooked kernels are the round-5 kernels: what the phase tables under
#     profiles/r0[3-5]_*phases* measured.
set -e
HOOKS_COMMIT=991d3f7
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
CS="$ROOT/acr_wsss_amd/csrc"
SRC="$CS"
if [ "$1" = "-H" ]; then
  shift
  SRC="$ROOT/scripts/lab/_build/hooks/acr_wsss_amd/csrc"
  mkdir -p "$ROOT/scripts/lab/_build/hooks"
  (cd "$ROOT" && git archive $HOOKS_COMMIT acr_wsss_amd/csrc include | tar -x -C scripts/lab/_build/hooks)
fi
name=$1; src=$2; flags=$3
mkdir -p "$ROOT/scripts/lab/_build"
make -C "$CS" -j8 > /dev/null
obj="$ROOT/scripts/lab/_build/${src%.hip}_$name.o"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function $flags -I"$SRC" -c "$SRC/$src" -o "$obj"
others=$(ls "$CS"/*.o | grep -v "/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/scripts/lab/_build/libacr_$name.so" $obj $others
echo "built scripts/lab/_build/libacr_$name.so"
