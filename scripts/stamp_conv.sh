#!/bin/bash
# Diagnostic: build libphendiff_hip with -DPD_STAMPS into /tmp, run one conv config, print phase breakdown.
set -e
HERE="$(cd "$(dirname "$0")/.." && pwd)"
cd "$HERE/phendiff_amd/csrc"
for f in conv_igemm attn_d8 small_kernels train_kernels backward_kernels wgrad sd_kernels; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DPD_STAMPS -c $f.hip -o /tmp/st_$f.o & done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libpd_stamps.so /tmp/st_conv_igemm.o /tmp/st_attn_d8.o /tmp/st_small_kernels.o /tmp/st_train_kernels.o /tmp/st_backward_kernels.o /tmp/st_wgrad.o /tmp/st_sd_kernels.o
cd "$HERE"
PD_LIB=/tmp/libpd_stamps.so python scripts/stamp_conv.py "$@"
