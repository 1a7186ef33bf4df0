#!/bin/bash
# Diagnostic: ONE -DPD_STAMPS build into /tmp, then the phase breakdown of several conv configurations ("--hw 256 --cin 64 ..." strings;
# a leading VAR=value word is exported for that run, e.g. "PD_CONV_PRO=1 --hw 256 --cin 64 --cout 64 --gn 1").
HERE="$(cd "$(dirname "$0")/.." && pwd)"
cd "$HERE/phendiff_amd/csrc"
SRCS="conv_igemm attn_d8 small_kernels train_kernels backward_kernels wgrad sd_kernels vae_kernels sd_bwd_kernels linear_gemm comm_rccl"
OBJS=""
for f in $SRCS; do
  X=""; { [ "$f" = "attn_d8" ] || [ "$f" = "sd_bwd_kernels" ]; } && X="-mllvm -amdgpu-mfma-vgpr-form"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DPD_STAMPS $EXTRA_HIPCC_FLAGS $X -c $f.hip -o /tmp/st_$f.o 2>/dev/null &
  OBJS="$OBJS /tmp/st_$f.o"
done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libpd_stamps.so $OBJS -ldl || exit 1
cd "$HERE"
for cfg in "$@"; do
  echo "=== $cfg"
  envs=""; rest=""
  for w in $cfg; do case "$w" in [A-Z]*=*) envs="$envs $w";; *) rest="$rest $w";; esac; done
  env $envs PD_LIB=/tmp/libpd_stamps.so PD_ALLOW_ABI_MISMATCH=1 python scripts/stamp_conv.py $rest 2>&1 | grep -v "occupancy API\|^device:"
done
