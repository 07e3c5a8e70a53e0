#!/bin/bash
# Round 5, experiment d: fft_quad_kernel with the LDS-DMA pieces spread over the epilogue; pinned constants on/off; priorities.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/r05d
mkdir -p $O
cd $ROOT
{
for v in pin nopin nopin_burst nopin_p21 nopin_p00 nopin nopin_tl; do
timeout 120 tools/ubench/bin/qb_$v 16 50 $v 4 | grep -v " ran"
done
timeout 120 tools/ubench/bin/qb_nopin 32 30 nopin32 4 | grep -v " ran"
QB_SPECIAL=1 QB_WARM=5 timeout 120 tools/ubench/bin/qb_nopin 16 4 nopin_special 4 | grep -v " ran"
} 2>&1 | tee $O/log.txt
