#!/bin/bash
# Builds libvfn.so (HIP kernels + C ABI) for gfx950, in-tree.  hipcc cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")"
ARCH=${VFN_ARCH:-gfx950}
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=${ARCH} -Wall -Wno-unused-function"
hipcc $FLAGS -c vfn_pack.hip -o vfn_pack.o &
hipcc $FLAGS -c vfn_mlp.hip -o vfn_mlp.o ${VFN_MLP_EXTRA:-} &
hipcc $FLAGS -c vfn_mlp_bwd.hip -o vfn_mlp_bwd.o &
hipcc $FLAGS -c vfn_dw16.hip -o vfn_dw16.o &
hipcc $FLAGS -c vfn_unfold.hip -o vfn_unfold.o &
hipcc $FLAGS -mllvm -amdgpu-mfma-vgpr-form -mllvm -pragma-unroll-threshold=10000000 -c vfn_bwd16.hip -o vfn_bwd16.o &
# vfn_mlp16: accumulators in arch VGPRs (all AGPRs hold activations), full unrolling of the K loops (see its header)
hipcc $FLAGS -mllvm -amdgpu-mfma-vgpr-form -mllvm -pragma-unroll-threshold=10000000 -c vfn_mlp16.hip -o vfn_mlp16.o ${VFN_MLP16_EXTRA:-} &
hipcc $FLAGS -ffp-contract=off -c vfn_rays.hip -o vfn_rays.o &
hipcc $FLAGS -ffp-contract=off -c vfn_grid.hip -o vfn_grid.o &
hipcc $FLAGS -c vfn_bstat.hip -o vfn_bstat.o &
wait
hipcc -shared -fPIC --offload-arch=${ARCH} -o libvfn.so vfn_pack.o vfn_mlp.o vfn_mlp_bwd.o vfn_dw16.o vfn_unfold.o vfn_bwd16.o vfn_mlp16.o vfn_rays.o vfn_grid.o vfn_bstat.o
echo "built $(pwd)/libvfn.so"
