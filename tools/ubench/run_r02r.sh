#!/bin/bash
# cache policy of the fused kernel's output stores: st<aux>, aux bits 1 = sc0, 2 = nt, 16 = sc1
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
B=$ROOT/tools/ubench/bin
O=$ROOT/gpurun_out/r02r
mkdir -p $O
for rep in 1 2 3; do
  for b in st0 st1 st2 st3 st16 st17 st18 st19; do timeout 120 $B/$b 300 $b 0 | grep "events"; done
done > $O/fb.log 2>&1
cat $O/fb.log
