#!/bin/bash
# round-3 batch J: the one-transform-per-workgroup kernel (4 workgroups per CU, no prefetch) re-measured with the lean fast epilogue
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
B=$ROOT/tools/ubench/bin
HC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -ffp-contract=off -D__HIP_PLATFORM_AMD__ -I $ROOT/cyberether_amd/csrc/kernels -I $ROOT/cyberether_amd/csrc -I $ROOT/include -I $ROOT/tools/ubench"
$HC -DFB_WG -DFB_FAST=true $ROOT/tools/ubench/fused_bench.hip -Rpass-analysis=kernel-resource-usage -o $B/j_wg_f 2>&1 | grep -A5 "fft_wg_kernel" | grep -E "VGPRs:|Scratch" | tr '\n' ' ' &
$HC -DFB_WG $ROOT/tools/ubench/fused_bench.hip -o $B/j_wg_e &
$HC -DFB_WG -DFB_TRIVIAL_EPI $ROOT/tools/ubench/fused_bench.hip -o $B/j_wg_t &
$HC -DFB_FAST=true $ROOT/tools/ubench/fused_bench.hip -o $B/j_p_f &
wait; echo; ls $B | grep -c "^j_"
