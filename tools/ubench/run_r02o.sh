#!/bin/bash
# exchange layout: two pads per 16 (q0 exact, q2 fast) vs one pad per 8 (q1, q3); bank conflicts by PMC
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
B=$ROOT/tools/ubench/bin
O=$ROOT/gpurun_out/r02o
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for rep in 1 2 3; do
  for b in q0 q1 q2 q3; do timeout 120 $B/$b 300 $b 0 | grep "events"; done
done > $O/fb.log 2>&1
for b in q0 q1; do
  timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU -d $O/pmc_$b -- $B/$b 40 $b 0 > $O/pmc_$b.log 2>&1
done
for d in $O/pmc_q*/; do echo "#### $d"; python3 $ROOT/tools/pmc_summary.py $d fft_pipe 2>&1 | tail -4; done > $O/pmc_summary.txt
cat $O/fb.log $O/pmc_summary.txt
