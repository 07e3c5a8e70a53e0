#!/bin/bash
# Round 5, q: SMALL launches (1, 2, 4 cycles = 1024..4096 transforms: what a live source gets): the 512-thread pipelined kernel
# (512 workgroups) against the quad kernel on 2, 3 and 4 workgroups per CU, static hand-out.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/r05q
mkdir -p $O
cd $ROOT
{
for c in 1 2 4 8; do
for g in 1 2 3 4; do
QB_STATIC=1 QB_WARM=100 timeout 120 tools/ubench/bin/qb_nosteal $c 200 "c${c}_g${g}" $g | grep "cycles per launch\|differing: [1-9]"
done
done
} 2>&1 | tee $O/log.txt
