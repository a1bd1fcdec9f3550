#!/bin/bash
# Ablation builds of three contraction kernels (wrong results, timing only): what their staging costs (DESIGN.md sections 3.2 / 8,
# CHANGELOG round 4).  Builds libvaehip_ablate_<kernel>_<bits>.so next to the library and times the layer shapes with it.
#   bits: 1 no global loads (the LDS-DMA kernel: every piece out of range, same instructions), 2 no LDS stores, 4 / 8 the wide-tile
#   kernel's halo loads / stores, 16 no barrier.          usage (GPU box): bash tools/ablation_builds.sh [wide|wgrad16|wgrad32]
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; C=$R/vae-channel-dynamics_amd/csrc; O=$R/gpurun_out/ablation; mkdir -p $O
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -Xclang -target-feature -Xclang -packed-fp32-ops -Wno-unused-result"
OBJS=$(make -C $C -pn 2>/dev/null | sed -n 's/^OBJS = //p' | head -1)
run() {  # kernel source, bits, microbench command
  local src=$1 bits=$2; shift 2
  /opt/rocm/bin/hipcc $FLAGS -DVAE_ABLATE=$bits -c $C/$src.hip -o $O/${src}_$bits.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $(for o in $OBJS; do [ $o = $src.o ] || echo $C/$o; done) $O/${src}_$bits.o -o $C/libvaehip_ablate_${src}_$bits.so
  echo "== $src VAE_ABLATE=$bits"; VAEHIP_LIB=$C/libvaehip_ablate_${src}_$bits.so "$@" | grep -v amdgpu.ids
  rm -f $C/libvaehip_ablate_${src}_$bits.so
}
make -C $C > /dev/null
case ${1:-wide} in
  wide)    for b in 0 1 3 15 31; do run conv3_wide_bf16 $b python $R/tools/microbench_bf16.py c128 c256 c512 | grep -E "==|fwd_img |dgrad_img"; done ;;
  wgrad16) for b in 0 1 3 19; do run wgrad3_tile_bf16 $b python $R/tools/microbench_bf16.py c128 c256 c512 | grep -E "==|wgrad_img"; done
           for b in 1 3 19; do VAEHIP_NO_WGRAD_DMA=1 run wgrad3_tile_bf16 $b python $R/tools/microbench_bf16.py c256 | grep -E "==|wgrad_img"; done ;;
  wgrad32) for b in 0 1 3 19; do MB_ONLY=wgrad run wgrad3_wino $b python $R/tools/microbench_wino.py c128 c256 c512 | grep -E "==|wino  *wgrad "; done ;;
esac
