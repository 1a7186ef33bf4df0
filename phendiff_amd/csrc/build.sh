#!/bin/bash
# Build libphendiff_hip.so for gfx950 (MI355X) in-tree.  Cross-compiles without a GPU.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/../libphendiff_hip.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
# --clean: compile every object from its source (what __graft_entry__.build() does: a library that provably matches the tree)
if [ "$1" = "--clean" ]; then rm -rf "$HERE/build"; rm -f "$OUT"; fi
mkdir -p "$HERE/build"
pids=()
SRCS="conv_igemm attn_d8 small_kernels train_kernels backward_kernels wgrad sd_kernels vae_kernels sd_bwd_kernels linear_gemm"
for f in $SRCS; do
  if [ ! -f "$HERE/build/$f.o" ] || [ "$HERE/$f.hip" -nt "$HERE/build/$f.o" ] || [ "$HERE/pd_common.h" -nt "$HERE/build/$f.o" ] || [ "$HERE/pd_stage.h" -nt "$HERE/build/$f.o" ] || [ "$HERE/pd_d64.h" -nt "$HERE/build/$f.o" ] || [ "$HERE/pd_conv.h" -nt "$HERE/build/$f.o" ] \
     || [ "$HERE/../../include/phendiff_hip.h" -nt "$HERE/build/$f.o" ]; then
    X=""
    # attention (forward d = 8, backward d = 64): keep MFMA accumulators in VGPRs (the softmax / its derivative work on them;
    # AGPR form costs a copy per register -- d64 backward: 20.3 -> 17.8 ms per SD training step, dq kernel 2 -> 3 waves/SIMD)
    if [ "$f" = "attn_d8" ] || [ "$f" = "sd_bwd_kernels" ]; then X="-mllvm -amdgpu-mfma-vgpr-form"; fi
    $HIPCC $FLAGS $X $EXTRA_HIPCC_FLAGS -c "$HERE/$f.hip" -o "$HERE/build/$f.o" &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
OBJS=""
for f in $SRCS; do OBJS="$OBJS $HERE/build/$f.o"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT" $OBJS
# manifest: hash of every source the library was built from (bench.py / the profile collectors compare it with the tree)
python3 - "$HERE" "$OUT" <<'PY'
import hashlib, json, os, sys
here, out = sys.argv[1:3]
sys.path.insert(0, os.path.join(here, "..", ".."))
from phendiff_amd._lib import source_hash
json.dump({"sources_sha256": source_hash(), "library_sha256": hashlib.sha256(open(out, "rb").read()).hexdigest(),
           "flags": os.environ.get("EXTRA_HIPCC_FLAGS", "")}, open(out + ".manifest.json", "w"))
PY
echo "built $OUT"
