#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
B=$ROOT/tools/ubench/bin
O=$ROOT/gpurun_out/r03h
mkdir -p $O
for rep in 1 2 3; do for k in e f; do for v in c0r0 c1r0 c0r1 c1r1; do timeout 120 $B/h_${k}_$v 300 h_${k}_$v 0 | grep events; done; done; done > $O/no_calls.log 2>&1
for v in h_e_c1r0 h_e_c1r1 h_e_c0r0; do timeout 120 $B/$v 300 ${v}_mode1 1 | grep events; timeout 120 $B/$v 100 ${v}_mode2 2 | grep events; done >> $O/no_calls.log 2>&1
cat $O/no_calls.log
