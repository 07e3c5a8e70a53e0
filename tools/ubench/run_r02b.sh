#!/bin/bash
# round-2 experiment batch B: decomposition of the fused kernel's time + PMC counters + sweep sensitivity
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
B=$ROOT/tools/ubench/bin
O=$ROOT/gpurun_out/r02b
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for s in sweep_bad sweep_opt; do echo "== $s"; timeout 300 $B/$s; done > $O/sweeps.log 2>&1
for rep in 1 2; do
  for b in b0 b0s b1 b2 b3 b4 b6 b7 b8; do timeout 120 $B/$b 300 $b 0; done
  timeout 120 $B/b7 300 b7_noguard 0 512 0
  timeout 120 $B/b0 300 b0_grid256 0 256
  timeout 120 $B/b2 300 b2_grid256 0 256
  timeout 120 $B/b0 300 b0_grid1024 0 1024
done > $O/fb.log 2>&1
rocprofv3 -L > $O/counters_list.txt 2>&1
for b in b0 b2 b6 b7; do
  timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU -d $O/pmc1_$b -- $B/$b 40 $b 0 > $O/pmc1_$b.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc SQ_IFETCH SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_TRANS GRBM_GUI_ACTIVE -d $O/pmc2_$b -- $B/$b 40 $b 0 > $O/pmc2_$b.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE -d $O/pmc3_$b -- $B/$b 40 $b 0 > $O/pmc3_$b.log 2>&1
done
for d in $O/pmc*_b*/; do echo "#### $d"; python3 $ROOT/tools/pmc_summary.py $d fft_pipe 2>&1 | head -40; done > $O/pmc_summary.txt
cat $O/sweeps.log $O/fb.log; tail -3 $O/pmc*_b0.log
