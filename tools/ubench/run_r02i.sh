#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
B=$ROOT/tools/ubench/bin
O=$ROOT/gpurun_out/r02i
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
  for b in q0 q1 q2; do timeout 120 $B/$b 300 $b 0; done
done > $O/fb.log 2>&1
cat $O/fb.log
