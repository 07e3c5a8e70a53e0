#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
B=$ROOT/tools/ubench/bin
O=$ROOT/gpurun_out/r03j
mkdir -p $O
for rep in 1 2 3; do for v in j_p_f j_wg_f j_wg_e j_wg_t; do timeout 120 $B/$v 300 $v 0 | grep events; done; done > $O/wg.log 2>&1
cat $O/wg.log
