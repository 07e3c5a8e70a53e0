#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
B=$ROOT/tools/ubench/bin
O=$ROOT/gpurun_out/r03b
mkdir -p $O
cd $ROOT
for rep in 1 2; do timeout 300 $B/floor_bisect 300 "E"; timeout 300 $B/floor_bisect 300 "G"; timeout 300 $B/floor_bisect 300 "A3"; timeout 300 $B/floor_bisect 300 "B2"; done > $O/floor_bisect2.log 2>&1
for rep in 1 2 3; do
  for k in e f t k; do for v in r0l0 r1l0 r0l1 r1l1; do
    timeout 120 $B/x_${k}_$v 300 x_${k}_$v 0 | grep events
  done; done
done > $O/resident.log 2>&1
timeout 120 $B/fft_timeline_r03 > $O/timeline.log 2>&1
cat $O/floor_bisect2.log $O/resident.log; head -40 $O/timeline.log
timeout 900 python -m pytest tests/test_gpu_exact_sweep.py tests/test_gpu_fast_provider.py tests/test_gpu_surfaces.py tests/test_gpu_chain.py -m gpu -x -q > $O/pytest.log 2>&1; tail -15 $O/pytest.log
