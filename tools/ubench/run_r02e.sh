#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
B=$ROOT/tools/ubench/bin
O=$ROOT/gpurun_out/r02e
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
  for b in d0 p1 p2 p3 p4 p5 p6 p7 d7 p1f p4f; do timeout 120 $B/$b 300 $b 0; done
done > $O/fb.log 2>&1
timeout 120 $B/timeline_p2 > $O/timeline_p2.log 2>&1
cat $O/fb.log; grep -E "==|mean phase|device span" $O/timeline_p2.log
