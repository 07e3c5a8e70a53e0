#!/bin/bash
# round-3 batch H: no function calls in the fused kernel (cold paths inlined) x resident operand, exact and fast
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
B=$ROOT/tools/ubench/bin
mkdir -p $B
HC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -ffp-contract=off -D__HIP_PLATFORM_AMD__ -I $ROOT/cyberether_amd/csrc/kernels -I $ROOT/cyberether_amd/csrc -I $ROOT/include -I $ROOT/tools/ubench"
for c in 0 1; do for r in 0 1; do
  $HC -DJST_COLD_INLINE=$c -DJST_OPND_RESIDENT=$r $ROOT/tools/ubench/fused_bench.hip -Rpass-analysis=kernel-resource-usage -o $B/h_e_c${c}r${r} 2>&1 | grep -A4 "Name: _ZN3jst3dev15fft_pipe" | grep -E "VGPRs:|Scratch" | tr '\n' ' ' &
  $HC -DFB_FAST=true -DJST_FAST_COLD_INLINE=1 -DJST_COLD_INLINE=$c -DJST_OPND_RESIDENT=$r $ROOT/tools/ubench/fused_bench.hip -o $B/h_f_c${c}r${r} 2>&1 | grep error &
done; done
wait; echo
ls $B | grep -c "^h_[ef]_"
