#!/bin/bash
# Build libphendiff_hip.so from the kernel sources of a git revision into build_ab/<name>.so (same-box A/B against the tree's
# library through PD_LIB).  Runs here (hipcc cross-compiles); build_ab/ is git-ignored but travels with gpurun.
#   scripts/build_rev.sh <rev> <name> [extra hipcc flags]
set -e
REV=$1; NAME=$2; shift; shift
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
TMP=$(mktemp -d)
mkdir -p "$TMP/phendiff_amd/csrc" "$TMP/include" "$ROOT/build_ab"
for f in $(git -C "$ROOT" ls-tree --name-only "$REV" phendiff_amd/csrc/ | grep -E '\.(hip|h)$'); do git -C "$ROOT" show "$REV:$f" > "$TMP/$f"; done
git -C "$ROOT" show "$REV:include/phendiff_hip.h" > "$TMP/include/phendiff_hip.h"
cd "$TMP/phendiff_amd/csrc"
OBJS=""
for f in $(ls *.hip | sed "s/\.hip$//"); do
  X=""; { [ "$f" = "attn_d8" ] || [ "$f" = "sd_bwd_kernels" ]; } && X="-mllvm -amdgpu-mfma-vgpr-form"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function $X "$@" -c $f.hip -o $f.o 2>/dev/null &
  OBJS="$OBJS $f.o"
done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/build_ab/$NAME.so" $OBJS -ldl
rm -rf "$TMP"
echo "built build_ab/$NAME.so from $REV"
