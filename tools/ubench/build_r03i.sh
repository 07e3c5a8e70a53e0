#!/bin/bash
# round-3 batch I: wave-priority sets re-checked for the lean fast kernel (the round-2 choice was made for a 2x heavier epilogue)
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
B=$ROOT/tools/ubench/bin
mkdir -p $B
HC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -ffp-contract=off -D__HIP_PLATFORM_AMD__ -I $ROOT/cyberether_amd/csrc/kernels -I $ROOT/cyberether_amd/csrc -I $ROOT/include -I $ROOT/tools/ubench"
i=0
for p in "3 3 0 1" "3 3 0 0" "3 3 1 1" "2 3 0 1" "3 2 1 0" "3 3 2 2" "1 1 0 0" "3 3 1 2" "2 2 0 1" "3 3 0 2"; do
  set -- $p
  $HC -DFB_FAST=true -DJST_PRIO_PA=$1 -DJST_PRIO_PB=$2 -DJST_PRIO_EA=$3 -DJST_PRIO_EB=$4 $ROOT/tools/ubench/fused_bench.hip -o $B/i_f_$1$2$3$4 &
  i=$((i+1)); if [ $((i % 5)) -eq 0 ]; then wait; fi
done
$HC -DFB_FAST=true -DJST_NO_SETPRIO $ROOT/tools/ubench/fused_bench.hip -o $B/i_f_none &
$HC -DFB_FAST=true -DJST_LOAD16=1 $ROOT/tools/ubench/fused_bench.hip -o $B/i_f_l16 &
$HC -DFB_FAST=true -DJST_LOAD16=1 -DJST_OPND_RESIDENT=1 $ROOT/tools/ubench/fused_bench.hip -o $B/i_f_l16r1 &
wait
ls $B | grep -c "^i_f_"
