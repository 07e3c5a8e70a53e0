#!/bin/bash
# Round 4, experiment c: epilogue v2 with the unbiased exponent: the sweeps + fast-provider suites again, then SQ counters of
# the pipelined and the one-wavefront-per-transform kernel (two passes of eight counters each) from bench.py runs.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/r04c
mkdir -p $O
cd $ROOT
echo "== exhaustive sweeps + fast provider + chain tests"
timeout 1500 python -m pytest tests/test_gpu_exact_sweep.py tests/test_gpu_fast_provider.py tests/test_gpu_batch.py tests/test_gpu_chain.py tests/test_gpu_spectrogram_indices.py -q 2>&1 | tail -8
JST_FFT_KERNEL=wave timeout 600 python -m pytest tests/test_gpu_batch.py tests/test_gpu_fast_provider.py -q 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
for k in pipe wave; do
  B="python $ROOT/bench.py --no-cpu-baseline --no-alt --no-parity --no-host-fed --min-time 0.05"
  mkdir -p $O/$k
  JST_FFT_KERNEL=$k timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS -d $O/$k/pmc_sq -- $B > $O/$k/pmc_sq.log 2>&1
  JST_FFT_KERNEL=$k timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT -d $O/$k/pmc_sq2 -- $B > $O/$k/pmc_sq2.log 2>&1
  JST_FFT_KERNEL=$k timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_IFETCH SQ_WAVES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INSTS_SMEM SQ_WAIT_INST_ANY -d $O/$k/pmc_sq3 -- $B > $O/$k/pmc_sq3.log 2>&1
  python $ROOT/tools/pmc_summary.py $O/$k > $O/pmc_counters_$k.txt 2>&1
  find $O/$k -name "*.db" -delete; find $O/$k -name "*kernel_trace.csv" -size +5M -delete
done
grep -A12 "fft_pipe_kernel\|fft_wave4096" $O/pmc_counters_pipe.txt $O/pmc_counters_wave.txt | head -80
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u | tr '\n' ' ' > $O/sq_counter_names.txt
