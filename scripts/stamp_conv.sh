#!/bin/bash
# Diagnostic: build libphendiff_hip with -DPD_STAMPS into /tmp, run one conv config, print phase breakdown.
set -e
HERE="$(cd "$(dirname "$0")/.." && pwd)"
cd "$HERE/phendiff_amd/csrc"
SRCS="conv_igemm attn_d8 small_kernels train_kernels backward_kernels wgrad sd_kernels vae_kernels sd_bwd_kernels linear_gemm comm_rccl"
OBJS=""
for f in $SRCS; do
  X=""; { [ "$f" = "attn_d8" ] || [ "$f" = "sd_bwd_kernels" ]; } && X="-mllvm -amdgpu-mfma-vgpr-form"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DPD_STAMPS $X -c $f.hip -o /tmp/st_$f.o &
  OBJS="$OBJS /tmp/st_$f.o"
done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libpd_stamps.so $OBJS -ldl
cd "$HERE"
PD_LIB=/tmp/libpd_stamps.so python scripts/stamp_conv.py "$@"
