# config 5, 128 transforms per cycle, one launch per unit and cycle: kernel stats + HBM traffic + SQ / LDS counters per provider
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/${OUTDIR:-c5pmc}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in ${PROVIDERS:-fast}; do
  C="python $ROOT/tools/bench_c5_streams.py $v ${B:-128} ${FORM:-per_cycle}"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$v/trace -- $C > $O/$v.json 2> $O/$v.err
  python $ROOT/tools/kstats.py $O/$v/trace > $O/kstats_$v.txt 2>&1
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/$v/pmc_fetch -- $C > /dev/null 2>&1
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/$v/pmc_write -- $C > /dev/null 2>&1
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS -d $O/$v/pmc_sq -- $C > /dev/null 2>&1
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE -d $O/$v/pmc_lds -- $C > /dev/null 2>&1
  for p in pmc_fetch pmc_write pmc_sq pmc_lds; do python $ROOT/tools/pmc_kernel_means.py $O/$v/$p fft_tile; done > $O/pmc_$v.txt 2>&1
  echo "== $v"; cat $O/$v.json; head -4 $O/kstats_$v.txt; cat $O/pmc_$v.txt
  find $O/$v -name "*.db" -delete; find $O/$v -name "*kernel_trace.csv" -delete; find $O/$v -name "*counter_collection.csv" -delete
done
