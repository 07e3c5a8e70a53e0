#!/bin/bash
# Round 5, experiment c: as b, without the harness's own tail (special rows), timeline symbol set before the first launch.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/r05c
mkdir -p $O
cd $ROOT
{
timeout 120 tools/ubench/bin/qb_B4 16 50 B4 4
QB_SPECIAL=1 timeout 120 tools/ubench/bin/qb_B4 16 10 B4special 4
timeout 120 tools/ubench/bin/qb_B4tl 16 20 B4tl 4
for v in A3bar A3early; do
echo "== $v, quad kernel alone, 1 cycle"
QB_ONLY=b QB_WARM=0 timeout 60 tools/ubench/bin/qb_$v 1 2 $v 3
done
} 2>&1 | tee $O/log.txt
