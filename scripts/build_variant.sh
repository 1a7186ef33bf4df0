# build conv_igemm.hip with extra flags and link with the tree's other objects into build_ab/$1.so (the tree's objects must be current: build.sh)
set -e
NAME=$1; shift
cd /root/repo/phendiff_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function "$@" -c conv_igemm.hip -o /tmp/conv_$NAME.o 2>/dev/null
OBJS=""
for f in attn_d8 small_kernels train_kernels backward_kernels wgrad sd_kernels vae_kernels sd_bwd_kernels linear_gemm linear_p8 metric_kernels comm_rccl; do OBJS="$OBJS build/$f.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/build_ab/$NAME.so /tmp/conv_$NAME.o $OBJS -ldl
echo built $NAME
