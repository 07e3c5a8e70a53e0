#!/bin/bash
# Round 5, experiment a: fft_quad_kernel / fft_quadd_kernel (fft_quad.hh: 256 threads x 16 points, in-place exchange, four
# workgroups per CU) against the product's fft_pipe_kernel at the bench launch size.  Bit-compare, then alternate timings.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/r05a
mkdir -p $O
cd $ROOT
for v in A4 A3 B4 B4plain B4prio; do
  for cyc in 16 32; do
    g=4; [ $v = A3 ] && g=3; timeout 120 tools/ubench/bin/qb_$v $cyc 20 $v $g 2>&1 | tee -a $O/quad_ab.log
  done
done
