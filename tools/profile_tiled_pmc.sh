#!/bin/bash
# Diagnostic (via gpurun): HBM traffic and SQ / LDS counters of the LDS-tiled FFT kernels at config 3 (bit-exact FFT
# overlap-add chain) and config 5 (8 streams resident).  PMC passes use --kernel-trace only.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/${1:-tiled_pmc}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() {  # name, command
  local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name/trace -- "$@" > $O/$name.trace.log 2>&1
  python $ROOT/tools/kstats.py $O/$name/trace > $O/kernel_stats_$name.txt 2>&1
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/$name/pmc_fetch -- "$@" > /dev/null 2>&1
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/$name/pmc_write -- "$@" > /dev/null 2>&1
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS -d $O/$name/pmc_sq -- "$@" > /dev/null 2>&1
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE -d $O/$name/pmc_lds -- "$@" > /dev/null 2>&1
  for p in pmc_fetch pmc_write pmc_sq pmc_lds; do python $ROOT/tools/pmc_kernel_means.py $O/$name/$p fft_tile; done > $O/pmc_$name.txt 2>&1
  find $O/$name -name "*.db" -delete; find $O/$name -name "*kernel_trace.csv" -delete; find $O/$name -name "*counter_collection.csv" -delete
}
run c3 python $ROOT/tools/bench_configs.py C3
run c5s python $ROOT/tools/bench_c5_streams.py
cat $O/kernel_stats_c3.txt | head -12; cat $O/pmc_c3.txt
cat $O/kernel_stats_c5s.txt | head -12; cat $O/pmc_c5s.txt
