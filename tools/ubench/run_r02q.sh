#!/bin/bash
# compiler scheduling strategies for the fused kernel: s0 default, s1 max-ilp, s2 max-memory-clause, s3 iterative-ilp, s4 amdgpu RP trackers
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
B=$ROOT/tools/ubench/bin
O=$ROOT/gpurun_out/r02q
mkdir -p $O
for rep in 1 2 3; do
  for b in s0 s1 s2 s3 s4; do timeout 120 $B/$b 300 $b 0 | grep "events"; done
done > $O/fb.log 2>&1
cat $O/fb.log
