#!/bin/bash
# Two / three hardware queues with independent launches of the fused kernel: tail-under-ramp overlap?
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
B=$ROOT/tools/ubench/bin
O=$ROOT/gpurun_out/r02l
mkdir -p $O
for rep in 1 2; do
  for b in ms_exact ms_fast; do timeout 120 $B/$b 300 $b 0; done
done > $O/fb.log 2>&1
cat $O/fb.log
