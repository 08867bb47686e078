#!/bin/bash
# One translation unit of libvfn.so rebuilt with extra definitions, linked with the product build's other objects:
#     tools/build_unit_variant.sh NAME UNIT "-DFLAG ..."   ->  tools/micro/libvfn_NAME.so   (run csrc/build.sh first; use with VFN_LIB=...)
# (tools/micro/ travels to the GPU box; scratch/ does not)
set -euo pipefail
name=$1; unit=$2; defs=${3:-}
root=$(cd "$(dirname "$0")/.." && pwd)
obj=/tmp/vfn_variant_$name
rm -rf "$obj"; mkdir -p "$obj"
cp -p "$root"/vf_nerf_amd/csrc/*.o "$obj"/
upper=$(echo "${unit#vfn_}" | tr a-z A-Z)
env "VFN_${upper}_EXTRA=$defs" VFN_ONLY="$unit" VFN_OBJDIR="$obj" VFN_OUT="$root/tools/micro/libvfn_$name.so" bash "$root/vf_nerf_amd/csrc/build.sh"
