#!/bin/bash
# round-3 batch E: the role-split kernel (fft_split_kernel) against the pipelined kernel, same box
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
B=$ROOT/tools/ubench/bin
mkdir -p $B
HC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -ffp-contract=off -D__HIP_PLATFORM_AMD__ -I $ROOT/cyberether_amd/csrc/kernels -I $ROOT/cyberether_amd/csrc -I $ROOT/include -I $ROOT/tools/ubench"
$HC $ROOT/tools/ubench/fused_bench.hip -o $B/p_e &
$HC -DFB_FAST=true $ROOT/tools/ubench/fused_bench.hip -o $B/p_f &
$HC -DFB_TRIVIAL_EPI $ROOT/tools/ubench/fused_bench.hip -o $B/p_t &
$HC -DFB_SPLIT $ROOT/tools/ubench/fused_bench.hip -o $B/s_e &
$HC -DFB_SPLIT -DFB_FAST=true $ROOT/tools/ubench/fused_bench.hip -o $B/s_f &
$HC -DFB_SPLIT -DFB_TRIVIAL_EPI $ROOT/tools/ubench/fused_bench.hip -o $B/s_t &
wait
ls $B | grep -E "^[ps]_[eft]$" | wc -l
