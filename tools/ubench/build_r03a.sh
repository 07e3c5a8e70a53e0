#!/bin/bash
# round-3 batch A: floor bisect + wide-access variants of the fused 4096-point kernel (built here, run on the GPU box)
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
B=$ROOT/tools/ubench/bin
mkdir -p $B
HC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -ffp-contract=off -D__HIP_PLATFORM_AMD__ -I $ROOT/cyberether_amd/csrc/kernels -I $ROOT/cyberether_amd/csrc -I $ROOT/include"
$HC $ROOT/tools/ubench/floor_bisect.hip -o $B/floor_bisect &
for s in 0 1; do for l in 0 1; do
  W="-DJST_STORE16=$s -DJST_LOAD16=$l"
  $HC $W $ROOT/tools/ubench/fused_bench.hip -o $B/w_e_s${s}l${l} &
  $HC $W -DFB_FAST=true $ROOT/tools/ubench/fused_bench.hip -o $B/w_f_s${s}l${l} &
  $HC $W -DFB_TRIVIAL_EPI $ROOT/tools/ubench/fused_bench.hip -o $B/w_t_s${s}l${l} &
  $HC $W -DFB_TRIVIAL_EPI -DJST_FB_SKIP_PASSES $ROOT/tools/ubench/fused_bench.hip -o $B/w_k_s${s}l${l} &
  wait
done; done
wait
ls -la $B | grep -E "floor_bisect|w_[eftk]_" | wc -l
