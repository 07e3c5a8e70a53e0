#!/bin/bash
# wave-priority sets (passes old/young, epilogue old/young) re-checked after the single-read exchange
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
B=$ROOT/tools/ubench/bin
O=$ROOT/gpurun_out/r02n
mkdir -p $O
for rep in 1 2 3; do
  for b in pr0 pr1 pr2 pr3 pr4 pr5; do timeout 120 $B/$b 300 $b 0 | grep "events"; done
done > $O/fb.log 2>&1
cat $O/fb.log
