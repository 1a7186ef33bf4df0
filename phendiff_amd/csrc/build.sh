#!/bin/bash
# Build libphendiff_hip.so for gfx950 (MI355X) in-tree.  Cross-compiles without a GPU.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/../libphendiff_hip.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
# --clean: compile every object from its source (what __graft_entry__.build() does: a library that provably matches the tree)
if [ "$1" = "--clean" ]; then rm -rf "$HERE/build"; rm -f "$OUT" "$OUT.manifest.json"; fi
mkdir -p "$HERE/build"
pids=()
SRCS="conv_igemm attn_d8 small_kernels train_kernels backward_kernels wgrad sd_kernels vae_kernels sd_bwd_kernels linear_gemm linear_p8 metric_kernels comm_rccl"
for f in $SRCS; do
  X=""
  # attention (forward d = 8, backward d = 64): keep MFMA accumulators in VGPRs (the softmax / its derivative work on them;
  # AGPR form costs a copy per register -- d64 backward: 20.3 -> 17.8 ms per SD training step, dq kernel 2 -> 3 waves/SIMD)
  if [ "$f" = "attn_d8" ] || [ "$f" = "sd_bwd_kernels" ]; then X="-mllvm -amdgpu-mfma-vgpr-form"; fi
  # an object is current when the hash of everything it depends on (its .hip, EVERY header, this script, the flags) equals the
  # one recorded beside it -- no hand-kept header list, no timestamps
  want="$(python3 "$HERE/source_hash.py" "$f" "$FLAGS $X $EXTRA_HIPCC_FLAGS")"
  if [ ! -f "$HERE/build/$f.o" ] || [ "$(cat "$HERE/build/$f.sha" 2>/dev/null)" != "$want" ]; then
    rm -f "$HERE/build/$f.sha"
    ( $HIPCC $FLAGS $X $EXTRA_HIPCC_FLAGS -c "$HERE/$f.hip" -o "$HERE/build/$f.o" && echo "$want" > "$HERE/build/$f.sha" ) &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
OBJS=""
for f in $SRCS; do OBJS="$OBJS $HERE/build/$f.o"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT" $OBJS -ldl
# manifest: hash of every source the library was built from (bench.py / the profile collectors compare it with the tree).
# Every object above carries the hash of its own inputs, so the set linked here IS the tree's.  No package / torch import.
python3 - "$HERE" "$OUT" <<'PY'
import hashlib, json, os, sys
here, out = sys.argv[1:3]
sys.path.insert(0, here)
from source_hash import source_hash
json.dump({"sources_sha256": source_hash(), "library_sha256": hashlib.sha256(open(out, "rb").read()).hexdigest(),
           "flags": os.environ.get("EXTRA_HIPCC_FLAGS", "")}, open(out + ".manifest.json", "w"))
PY
echo "built $OUT"
