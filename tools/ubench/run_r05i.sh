#!/bin/bash
# Round 5, i: static rounds first, claimed transforms for the rest of the launch (1/2, 3/4, 7/8 static) against all-static.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/r05i
mkdir -p $O
cd $ROOT
{
for rep in 1 2; do
for v in dyn12 dyn34 dyn78; do
timeout 120 tools/ubench/bin/qb_$v 16 50 $v 4 | grep "cycles per launch\|differing"
done
QB_STATIC=1 timeout 120 tools/ubench/bin/qb_dyn34 16 50 static 4 | grep "cycles per launch\|differing"
done
timeout 120 tools/ubench/bin/qb_dyn34 32 30 dyn34_32 4 | grep "cycles per launch\|differing"
QB_STATIC=1 timeout 120 tools/ubench/bin/qb_dyn34 32 30 static_32 4 | grep "cycles per launch\|differing"
timeout 120 tools/ubench/bin/qb_dyn34 1 100 dyn34_1 4 | grep "cycles per launch\|differing"
QB_SPECIAL=1 QB_WARM=5 timeout 120 tools/ubench/bin/qb_dyn34 16 10 dyn34_special 4 | grep "cycles per launch\|differing"
timeout 120 tools/ubench/bin/qb_dyn_tl 16 10 dyn34_tl 4 | grep "timeline\|workgroup"
} 2>&1 | tee $O/log.txt
