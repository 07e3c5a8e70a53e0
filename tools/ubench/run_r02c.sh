#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
B=$ROOT/tools/ubench/bin
O=$ROOT/gpurun_out/r02c
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
  for b in b6 w0 w1 w2 w7 w9; do timeout 120 $B/$b 300 $b 0; done
  timeout 120 $B/w7 300 w7_noguard 0 1024 0
  timeout 120 $B/w0 300 w0 1
done > $O/fb.log 2>&1
for b in w0 w2; do
  timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU -d $O/pmc1_$b -- $B/$b 40 $b 0 > $O/pmc1_$b.log 2>&1
done
for d in $O/pmc*_*/; do echo "#### $d"; python3 $ROOT/tools/pmc_summary.py $d fft_ 2>&1 | head -40; done > $O/pmc_summary.txt
cat $O/fb.log $O/pmc_summary.txt
