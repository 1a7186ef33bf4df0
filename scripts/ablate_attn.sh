#!/bin/bash
# Diagnostic: build attn_d8.hip with extra -D flags into /tmp and time the attention kernel with it.
set -e
HERE="$(cd "$(dirname "$0")/.." && pwd)"; DEFS="$1"; shift
cd "$HERE/phendiff_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form $DEFS -c attn_d8.hip -o /tmp/ab_attn_d8.o
for f in conv_igemm small_kernels train_kernels backward_kernels wgrad sd_kernels; do [ -f /tmp/ab_$f.o ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c $f.hip -o /tmp/ab_$f.o; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libpd_abl.so /tmp/ab_conv_igemm.o /tmp/ab_attn_d8.o /tmp/ab_small_kernels.o /tmp/ab_train_kernels.o /tmp/ab_backward_kernels.o /tmp/ab_wgrad.o /tmp/ab_sd_kernels.o
cd "$HERE"; PD_LIB=/tmp/libpd_abl.so python scripts/bench_attn.py "$@"
