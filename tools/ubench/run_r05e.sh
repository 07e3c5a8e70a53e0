#!/bin/bash
# Round 5, experiment e: both butterflies' LDS operands requested ahead in every pass (ra_*) against the previous form (old_*).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/r05e
mkdir -p $O
cd $ROOT
{
for rep in 1 2 3; do
for v in old_pin ra_pin old_nopin ra_nopin; do
timeout 120 tools/ubench/bin/qb_$v 16 50 $v 4 | grep "cycles per launch\|differing"
done
done
timeout 120 tools/ubench/bin/qb_ra_nopin_tl 16 20 ra_nopin_tl 4 | grep timeline
} 2>&1 | tee $O/log.txt
