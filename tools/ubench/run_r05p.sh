#!/bin/bash
# Round 5, p: a workgroup whose claim counter has run out tries the partner counters (b ^ 4, ^ 2, ^ 1) against stopping there.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/r05p
mkdir -p $O
cd $ROOT
{
for rep in 1 2 3; do
for v in steal nosteal; do
timeout 120 tools/ubench/bin/qb_$v 16 50 $v 4 | grep "cycles per launch\|differing: [1-9]"
done
done
timeout 120 tools/ubench/bin/qb_steal 32 30 steal32 4 | grep "cycles per launch\|differing: [1-9]"
timeout 120 tools/ubench/bin/qb_nosteal 32 30 nosteal32 4 | grep "cycles per launch\|differing: [1-9]"
timeout 120 tools/ubench/bin/qb_steal_tl 16 10 steal_tl 4 | grep "workgroup"
} 2>&1 | tee $O/log.txt
