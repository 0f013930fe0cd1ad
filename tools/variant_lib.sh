#!/usr/bin/env bash
# Developer aid: an A/B library that differs from the in-tree build in ONE translation unit compiled with extra flags.
#   tools/variant_lib.sh <name> <unit> [flags...]   ->  build/lib_<name>.so   (load it with QPN_HIP_LIB=build/lib_<name>.so)
# The other objects are taken from csrc/_obj (run csrc/build.sh first).
set -euo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
name="$1"; unit="$2"; shift 2
src="$root/quadraticprogramnetworks.jl_amd/csrc"
mkdir -p "$root/build/obj_$name"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-parameter "$@" -c "$src/$unit.hip" -o "$root/build/obj_$name/$unit.o"
objs=()
for o in "$src"/_obj/*.o; do b="$(basename "$o")"; if [[ "$b" == "$unit.o" ]]; then objs+=("$root/build/obj_$name/$unit.o"); else objs+=("$o"); fi; done
hipcc --offload-arch=gfx950 -fPIC -shared -o "$root/build/lib_$name.so" "${objs[@]}"
echo "built build/lib_$name.so"
