#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
B=$ROOT/tools/ubench/bin
O=$ROOT/gpurun_out/r02h
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
  for b in e_nopairs e0 e1 e1_nopairs e7 e7_noslp; do timeout 120 $B/$b 300 $b 0; done
  timeout 120 $B/e0 300 e0 1
done > $O/fb.log 2>&1
cat $O/fb.log
