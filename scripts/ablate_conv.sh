#!/bin/bash
# Diagnostic: build the library with extra -D flags into /tmp and time one conv config with it.
#   scripts/ablate_conv.sh "-DPD_ABL_W0" --hw 256 --cin 64 --cout 64
set -e
HERE="$(cd "$(dirname "$0")/.." && pwd)"
DEFS="$1"; shift
cd "$HERE/phendiff_amd/csrc"
for f in conv_igemm attn_d8 small_kernels train_kernels backward_kernels wgrad sd_kernels; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $DEFS -c $f.hip -o /tmp/ab_$f.o & done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libpd_abl.so /tmp/ab_conv_igemm.o /tmp/ab_attn_d8.o /tmp/ab_small_kernels.o /tmp/ab_train_kernels.o /tmp/ab_backward_kernels.o /tmp/ab_wgrad.o /tmp/ab_sd_kernels.o
cd "$HERE"
python - "$@" <<'PY'
import os, sys, runpy
sys.path.insert(0, os.getcwd())
import phendiff_amd._lib as L
L.LIB_PATH = "/tmp/libpd_abl.so"
sys.argv = ["bench_conv.py"] + sys.argv[1:]
runpy.run_path("scripts/bench_conv.py", run_name="__main__")
PY
