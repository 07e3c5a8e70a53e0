#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
B=$ROOT/tools/ubench/bin
O=$ROOT/gpurun_out/r02j
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
  for b in m0 m1 m2 m3; do timeout 120 $B/$b 300 $b 0; done
done > $O/fb.log 2>&1
for b in m0 m1; do
timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/pmcf_$b -- $B/$b 40 $b 0 > $O/pmcf_$b.log 2>&1
timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/pmcw_$b -- $B/$b 40 $b 0 > $O/pmcw_$b.log 2>&1
done
for d in $O/pmc*_m*/; do echo "#### $d"; python3 $ROOT/tools/pmc_summary.py $d fft_pipe 2>&1 | tail -2; done > $O/pmc_summary.txt
cat $O/fb.log $O/pmc_summary.txt
