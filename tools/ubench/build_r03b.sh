#!/bin/bash
# round-3 batch B: resident window operand x 16-byte loads on the fused kernel (exact / lean fast / trivial epilogue /
# skeleton), the per-workgroup timeline of the current kernel, and the second half of the floor bisect
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
B=$ROOT/tools/ubench/bin
mkdir -p $B
HC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -ffp-contract=off -D__HIP_PLATFORM_AMD__ -I $ROOT/cyberether_amd/csrc/kernels -I $ROOT/cyberether_amd/csrc -I $ROOT/include"
$HC $ROOT/tools/ubench/floor_bisect.hip -o $B/floor_bisect &
$HC -DJST_FFT_TIMELINE $ROOT/tools/ubench/fft_timeline.hip -o $B/fft_timeline_r03 &
for r in 0 1; do for l in 0 1; do
  W="-DJST_STORE16=0 -DJST_LOAD16=$l -DJST_OPND_RESIDENT=$r"
  $HC $W $ROOT/tools/ubench/fused_bench.hip -o $B/x_e_r${r}l${l} &
  $HC $W -DFB_FAST=true $ROOT/tools/ubench/fused_bench.hip -o $B/x_f_r${r}l${l} &
  $HC $W -DFB_TRIVIAL_EPI $ROOT/tools/ubench/fused_bench.hip -o $B/x_t_r${r}l${l} &
  $HC $W -DFB_TRIVIAL_EPI -DJST_FB_SKIP_PASSES $ROOT/tools/ubench/fused_bench.hip -o $B/x_k_r${r}l${l} &
  wait
done; done
wait
ls $B | grep -E "^x_|floor_bisect|fft_timeline_r03" | wc -l
