#!/bin/bash
# single ds_read_b64 (n0 exact, n2 fast) vs compiler-paired ds_read2_b64 (n1, n3) in the exchange reads
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
B=$ROOT/tools/ubench/bin
O=$ROOT/gpurun_out/r02m
mkdir -p $O
for rep in 1 2 3; do
  for b in n0 n4 n2 n5; do timeout 120 $B/$b 300 $b 0 | grep "events"; done
done > $O/fb.log 2>&1
cat $O/fb.log
