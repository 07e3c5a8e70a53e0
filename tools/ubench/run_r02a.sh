#!/bin/bash
# round-2 experiment batch A: exhaustive sweeps, A/B of epilogue + twiddle forms, timelines
cd "$(dirname "$0")/bin"
mkdir -p ../../../gpurun_out/r02a
O=../../../gpurun_out/r02a
for s in sweep_full sweep_div_v1_div_v1 sweep_div_v1_div_rn_midrange sweep_div_rn_midrange_div_v1 sweep_div_v2_div_v2 sweep_div_v3_div_v3; do
  echo "== $s"; timeout 300 ./$s
done > $O/sweeps.log 2>&1
for rep in 1 2; do
for b in fb_old fb_epi fb_tw fb_new fb_new_v1 fb_fast_old fb_fast_new; do
  timeout 120 ./$b 300 $b 0
done
done > $O/fb.log 2>&1
for b in fb_old fb_new fb_new_v1; do timeout 120 ./$b 200 $b 1; timeout 120 ./$b 200 $b 2; done >> $O/fb.log 2>&1
timeout 120 ./fft_timeline_old > $O/timeline_old.log 2>&1
timeout 120 ./fft_timeline_new > $O/timeline_new.log 2>&1
cat $O/sweeps.log $O/fb.log
