#!/bin/bash
# build_variant.sh NAME "<extra hipcc flags>": libjetstream_hip.so with fft_side.hip + fft_kernels.hip (or $VARIANT_UNITS) recompiled under
# the extra flags, as cyberether_amd/lib/variants/NAME.so (same-box A/B: the run script copies it over the library).
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
NAME=$1; shift
V=$ROOT/cyberether_amd/lib/variants; mkdir -p $V/obj_$NAME
FLAGS="-O3 -std=c++20 -fPIC -ffp-contract=off -fvisibility=hidden --offload-arch=gfx950 -Wno-unused-function -Wno-unused-result -D__HIP_PLATFORM_AMD__"
cd $ROOT/cyberether_amd/csrc
UNITS=${VARIANT_UNITS:-"fft_side fft_kernels"}   # translation units recompiled under the extra flags
for f in $UNITS; do
  /opt/rocm/bin/hipcc $FLAGS "$@" -c kernels/$f.hip -o $V/obj_$NAME/$f.o 2>/dev/null &
done
wait
PAT=$(echo $UNITS | tr ' ' '|')
OBJS=$(find ../lib/obj -name '*.o' | grep -v -E "kernels/($PAT)\.o")
NEW=$(for f in $UNITS; do echo $V/obj_$NAME/$f.o; done)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $V/$NAME.so $OBJS $NEW && echo "built $V/$NAME.so"
