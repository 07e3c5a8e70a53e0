#!/bin/bash
# Round 5, experiment b: warmed-up timings of fft_quadd_kernel (B4), its phase timeline, and fault isolation of variant A.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/r05b
mkdir -p $O
cd $ROOT
{
timeout 120 tools/ubench/bin/qb_B4 16 50 B4 4
timeout 120 tools/ubench/bin/qb_B4tl 16 20 B4tl 4
echo "== A3, quad kernel alone, 1 cycle"
QB_ONLY=b QB_WARM=0 timeout 60 tools/ubench/bin/qb_A3 1 2 A3only 3
echo "== A3, pipe kernel alone, 1 cycle"
QB_ONLY=a QB_WARM=0 timeout 60 tools/ubench/bin/qb_A3 1 2 A3pipe 3
} 2>&1 | tee $O/log.txt
