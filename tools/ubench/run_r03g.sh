#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
B=$ROOT/tools/ubench/bin
O=$ROOT/gpurun_out/r03g
mkdir -p $O
for rep in 1 2 3; do for v in g_f_c0r0 g_f_c1r0 g_f_c0r1 g_f_c1r1; do timeout 120 $B/$v 300 $v 0 | grep events; done; done > $O/cold_inline.log 2>&1
for v in g_f_c0r0 g_f_c1r0 g_f_c1r1; do timeout 120 $B/$v 300 ${v}_mode1 1 | grep events; done >> $O/cold_inline.log 2>&1
cat $O/cold_inline.log
