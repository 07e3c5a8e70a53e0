#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
B=$ROOT/tools/ubench/bin
O=$ROOT/gpurun_out/r03i
mkdir -p $O
for rep in 1 2 3; do for v in $(ls $B | grep "^i_f_"); do timeout 120 $B/$v 300 $v 0 | grep events; done; done > $O/prio.log 2>&1
cat $O/prio.log
