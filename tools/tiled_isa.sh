#!/bin/bash
# tiled_isa.sh <mangled-name regex>: device ISA of fft_tiled.hip, the first kernel whose label matches; prints the memory / wait
# skeleton and the register counts.  Diagnostic.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT/cyberether_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++20 -fPIC -ffp-contract=off -fvisibility=hidden --offload-arch=gfx950 -D__HIP_PLATFORM_AMD__ $EXTRA --cuda-device-only -S kernels/${UNIT:-fft_tiled}.hip -o /tmp/unit.s 2>&1 | grep error -A5
L=$(grep -n "^_ZN.*$1.*:" /tmp/unit.s | head -1 | cut -d: -f1)
awk -v s=$L 'NR>=s' /tmp/unit.s > /tmp/k.s
END=$(grep -n "s_endpgm" /tmp/k.s | head -1 | cut -d: -f1)
head -n $END /tmp/k.s > /tmp/k1.s
wc -l /tmp/k1.s
grep -n "s_waitcnt vmcnt\|s_barrier\|buffer_load\|global_load\|buffer_store\|global_store\|scratch_\|s_cbranch" /tmp/k1.s | head -${LINES_MAX:-120}
grep -n "\.vgpr_count\|scratch_en\|private_segment_fixed_size" /tmp/k.s | head -4
