#!/bin/bash
# Builds variants of the f16x3 kernel next to the product library: vf_nerf_amd/csrc/libvfn_<NAME>.so
#   tools/build_variants.sh "nodma:-DABL_NODMA" "fd3:-DVFN16_FDEPTH=3"      (run build.sh first: the other objects are reused)
# Switches understood by vfn_mlp16.hip: ABL_NODMA / ABL_NOEPI / ABL_NOSYNC (timing-only: results are wrong),
# VFN16_FDEPTH=n, VFN16_EPI_PER_MFMA=n, VFN16_NOGROUPS, VFN16_NOSCHED, VFN16_ASCALE=0, VFN16_STAMPS (timing-only).
set -euo pipefail
cd "$(dirname "$0")/../vf_nerf_amd/csrc"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form -mllvm -pragma-unroll-threshold=10000000"
for v in "$@"; do
  name=${v%%:*}; defs=${v#*:}
  ( hipcc $FLAGS $defs -c vfn_mlp16.hip -o /tmp/vfn_mlp16_$name.o && \
    hipcc -shared -fPIC --offload-arch=gfx950 -o libvfn_$name.so vfn_pack.o vfn_mlp.o vfn_mlp_bwd.o vfn_dw16.o vfn_dwf.o vfn_unfold.o vfn_bwd16.o /tmp/vfn_mlp16_$name.o vfn_rays.o vfn_grid.o vfn_bstat.o vfn_adam.o vfn_render.o vfn_wgrad.o && echo built $name ) &
done
wait
