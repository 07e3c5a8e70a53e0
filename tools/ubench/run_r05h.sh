#!/bin/bash
# Round 5, h: dynamic hand-out of transforms (device counter) against the static round robin, same binary.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/r05h
mkdir -p $O
cd $ROOT
{
for rep in 1 2 3; do
timeout 120 tools/ubench/bin/qb_dyn 16 50 dynamic 4 | grep "cycles per launch\|differing"
QB_STATIC=1 timeout 120 tools/ubench/bin/qb_dyn 16 50 static 4 | grep "cycles per launch\|differing"
done
timeout 120 tools/ubench/bin/qb_dyn 32 30 dynamic32 4 | grep "cycles per launch\|differing"
timeout 120 tools/ubench/bin/qb_dyn 1 100 dynamic1 4 | grep "cycles per launch\|differing"
QB_STATIC=1 timeout 120 tools/ubench/bin/qb_dyn 1 100 static1 4 | grep "cycles per launch\|differing"
QB_SPECIAL=1 QB_WARM=5 timeout 120 tools/ubench/bin/qb_dyn 16 10 dynamic_special 4 | grep "cycles per launch\|differing"
timeout 120 tools/ubench/bin/qb_dyn_tl 16 10 dyn_tl 4 | grep "timeline\|workgroup"
QB_STATIC=1 timeout 120 tools/ubench/bin/qb_dyn_tl 16 10 static_tl 4 | grep "timeline\|workgroup"
} 2>&1 | tee $O/log.txt
