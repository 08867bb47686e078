#!/bin/bash
# Builds libvfn.so (HIP kernels + C ABI) for gfx950, in-tree.  hipcc cross-compiles without a GPU.
# Every object is rebuilt from its source on every run (stale objects are deleted first) and every compile job's exit
# status is checked one by one: a failing translation unit fails the build, it can never be linked from an older .o.
set -uo pipefail
cd "$(dirname "$0")"
ARCH=${VFN_ARCH:-gfx950}
OUT=${VFN_OUT:-libvfn.so}
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=${ARCH} -Wall -Wno-unused-function"
# vfn_mlp16 / vfn_bwd16: accumulators in arch VGPRs (all AGPRs hold activations), full unrolling of the K loops
MFMA16="-mllvm -amdgpu-mfma-vgpr-form -mllvm -pragma-unroll-threshold=10000000"
UNITS="vfn_pack vfn_mlp vfn_mlp_bwd vfn_dw16 vfn_dwf vfn_unfold vfn_bwd16 vfn_mlp16 vfn_rays vfn_grid vfn_bstat vfn_adam vfn_render vfn_wgrad vfn_loss vfn_train"

extra_flags() {
  case "$1" in
    vfn_mlp)    echo "${VFN_MLP_EXTRA:-}" ;;
    vfn_mlp16)  echo "$MFMA16 ${VFN_MLP16_EXTRA:-}" ;;
    vfn_bwd16)  echo "$MFMA16 ${VFN_BWD16_EXTRA:-}" ;;
    vfn_dw16)   echo "${VFN_DW16_EXTRA:-}" ;;
    vfn_rays|vfn_grid) echo "-ffp-contract=off" ;;
    *) v="VFN_$(echo "${1#vfn_}" | tr a-z A-Z)_EXTRA"; echo "${!v:-}" ;;      # (any other unit: VFN_<UNIT>_EXTRA, e.g. tools/build_unit_variant.sh)
  esac
}

OBJDIR=${VFN_OBJDIR:-.}
mkdir -p "$OBJDIR"
rm -f "$OUT.tmp"
declare -A PIDS
OBJS=""
# VFN_ONLY="unit unit": developer shortcut — recompile only these units and link them with the other units' EXISTING objects (which
# must be newer than their sources; anything else is refused).  The default, and what __graft_entry__.build() runs, rebuilds everything.
ONLY=${VFN_ONLY:-}
for u in $UNITS; do
  [ -f "$u.hip" ] || { echo "build.sh: missing source $u.hip" >&2; exit 1; }
  if [ -n "$ONLY" ] && ! echo " $ONLY " | grep -q " $u "; then
    if [ -s "$OBJDIR/$u.o" ] && [ "$OBJDIR/$u.o" -nt "$u.hip" ] && [ "$OBJDIR/$u.o" -nt vfn_common.h ] && [ "$OBJDIR/$u.o" -nt ../../include/vfn.h ]; then
      OBJS="$OBJS $OBJDIR/$u.o"; touch "$OBJDIR/$u.remarks"; continue
    fi
    echo "build.sh: VFN_ONLY given but $u.o is missing or older than its sources" >&2; exit 1
  fi
  rm -f "$OBJDIR/$u.o"
  # shellcheck disable=SC2046
  hipcc $FLAGS $(extra_flags "$u") -Rpass-analysis=kernel-resource-usage -c "$u.hip" -o "$OBJDIR/$u.o" 2> "$OBJDIR/$u.remarks" &
  PIDS[$u]=$!
  OBJS="$OBJS $OBJDIR/$u.o"
done
failed=""
for u in $UNITS; do
  [ -n "${PIDS[$u]:-}" ] || continue
  if ! wait "${PIDS[$u]}"; then failed="$failed $u.hip"; fi
done
if [ -n "$failed" ]; then
  for f in $failed; do grep -E "error|Error" "$OBJDIR/${f%.hip}.remarks" | head -20 >&2; done
  echo "build.sh: compile FAILED for:$failed" >&2
  rm -f "$OUT"
  exit 1
fi
# The hand-pipelined one-wave-per-SIMD kernels live on the edge of the register file: a spill (scratch) does not break them,
# it makes them 1.6x slower without any other symptom (it happened: one loop-carried VGPR in the f16x3 epilogue).  Fail the build.
spilled=$(python3 - "$OBJDIR" <<'PY'
import re, sys, os
bad = []
for unit in ("vfn_mlp16", "vfn_bwd16", "vfn_dwf", "vfn_dw16", "vfn_bstat"):
    path = os.path.join(sys.argv[1], unit + ".remarks")
    name = None
    for line in open(path, errors="replace"):
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
        if m and name and "_kernel" in name and "pack" not in name and int(m.group(1)) > 0:
            bad.append(f"{unit}: {name} uses {m.group(1)} bytes/lane of scratch")
print("\n".join(bad))
PY
)
if [ -n "$spilled" ]; then
  echo "build.sh: register spills in a hot kernel:" >&2
  echo "$spilled" >&2
  rm -f "$OUT"
  exit 1
fi
for u in $UNITS; do
  [ -s "$OBJDIR/$u.o" ] || { echo "build.sh: $u.o missing after compile" >&2; rm -f "$OUT"; exit 1; }
done
# shellcheck disable=SC2086
hipcc -shared -fPIC --offload-arch=${ARCH} -o "$OUT.tmp" $OBJS || { rm -f "$OUT" "$OUT.tmp"; exit 1; }
mv -f "$OUT.tmp" "$OUT"
# the binding generates its signatures from the header: leave a copy inside the package for installs without the repository around them
cp -f ../../include/vfn.h ../vfn_abi.h
echo "built $(pwd)/$OUT"
