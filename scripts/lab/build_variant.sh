#!/bin/bash
# Lab build of the library with one source recompiled under extra flags:
#   scripts/lab/build_variant.sh <name> <file.hip> "<flags>"   ->  scripts/lab/_build/libacr_<name>.so  (use with ACR_LIB_PATH)
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
CS="$ROOT/acr_wsss_amd/csrc"
name=$1; src=$2; flags=$3
mkdir -p "$ROOT/scripts/lab/_build"
make -C "$CS" -j8 > /dev/null
obj="$ROOT/scripts/lab/_build/${src%.hip}_$name.o"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function $flags -I"$CS" -c "$CS/$src" -o "$obj"
others=$(ls "$CS"/*.o | grep -v "/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/scripts/lab/_build/libacr_$name.so" $obj $others
echo "built scripts/lab/_build/libacr_$name.so"
