#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
B=$ROOT/tools/ubench/bin
O=$ROOT/gpurun_out/r02d
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
  for b in b6 d0 d1 d2 d7 d7s; do timeout 120 $B/$b 300 $b 0; done
  timeout 120 $B/d7 300 d7_noguard 0 512 0
  timeout 120 $B/d0 300 d0 1
done > $O/fb.log 2>&1
timeout 120 $B/timeline_d0 > $O/timeline_d0.log 2>&1
timeout 300 $B/sweep_default > $O/sweep_default.log 2>&1
cat $O/fb.log; tail -3 $O/sweep_default.log; grep -E "==|mean phase|device span" $O/timeline_d0.log
